import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
graph = G.make_graph("ogbn-arxiv", 1500)
base = ["--dataset", "ogbn-arxiv", "--n-hidden", "128", "--n-classes", "10", "--bit_width", "2", "--use_QGTC", "--quiet", "--n-epochs", "20"]
res = {}
for name, extra in (("plain", []), ("streams1", ["--streams", "1"]), ("streams2", ["--streams", "2"]), ("streams4", ["--streams", "4"]), ("graph", ["--graph"]), ("batched", ["--batched"])):
    args = driver.build_parser().parse_args(base + extra)
    driver.run(args, Q=Q, graph=graph)
    r = driver.run(args, Q=Q, graph=graph)
    res[name] = r
    print(name, round(r["avg_epoch_ms"], 3), "ms/epoch")
for i in range(len(res["plain"]["outs"])):
    assert torch.equal(res["plain"]["outs"][i], res["graph"]["outs"][i])
    assert torch.equal(res["plain"]["outs"][i], res["batched"]["outs"][i])
    assert torch.equal(res["plain"]["outs"][i], res["streams4"]["outs"][i])
print("outputs identical across plain / graph / batched")
