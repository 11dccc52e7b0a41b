"""Round 6: what the reference's checked-in script launches (0_7a_eval_QGTC_cluster_GCN.py:6-10: --bit_width 32, hidden 16) on the
ogbn-arxiv-sized graph: Avg. Epoch of the unchanged per-batch loop and of the grouped plans (benchmarks/epochs.py's leg alone)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import QGTC as Q  # noqa: E402
from benchmarks import epochs  # noqa: E402

print(json.dumps(epochs.checked_in_script_settings(Q, 0, 1, 0), indent=1))
