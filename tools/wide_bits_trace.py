"""Per-kernel time of the unchanged driver's per-batch loop at the reference's checked-in setting (0_7a_eval_QGTC_cluster_GCN.py:6-10:
--bit_width 32, hidden 16) - run under `rocprofv3 --kernel-trace --stats`.  usage: wide_bits_trace.py [bits] [epochs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

bits = sys.argv[1] if len(sys.argv) > 1 else "32"
g = G.make_graph("ogbn-arxiv", 1500)
args = driver.build_parser().parse_args(["--dataset", "ogbn-arxiv", "--n-hidden", "16", "--n-classes", "10", "--bit_width", bits, "--use_QGTC", "--quiet",
                                         "--n-epochs", sys.argv[2] if len(sys.argv) > 2 else "2"])
it = driver.make_iter(args, Q, g)
print(driver.run(args, Q=Q, graph=g, it=it)["avg_epoch_ms"])
