"""The grouped layout-correct Cluster-GCN epoch (ogbn-arxiv-sized) at a width / hidden size, three runs of 20 epochs - for
`rocprofv3 --kernel-trace --stats`.  usage: epoch_trace.py bits [hidden]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
bits = sys.argv[1]
hidden = sys.argv[2] if len(sys.argv) > 2 else "128"
g = G.make_graph("ogbn-arxiv", 1500)
args = driver.build_parser().parse_args(["--dataset", "ogbn-arxiv", "--n-hidden", hidden, "--n-classes", "10", "--bit_width", bits, "--use_QGTC", "--quiet", "--n-epochs", "20", "--batched", "--chain", "correct"])
it = driver.make_iter(args, Q, g)
for _ in range(3): r = driver.run(args, Q=Q, graph=g, it=it)
print(r["avg_epoch_ms"])
