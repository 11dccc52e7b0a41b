import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
print("cpus", os.cpu_count(), "affinity before", len(os.sched_getaffinity(0)))
order = sys.argv[1]
from oracle.qgtc_oracle import Oracle
def bench_oracle(tag):
    O = Oracle()
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, size=(600, 600)).astype(np.int32)
    X = O.pack(q, 8, False); W = O.pack(q, 8, True)
    O.bitmm2bit(X, W, 600, 600, 600, 8, 8, 8)
    t = time.time(); O.bitmm2bit(X, W, 600, 600, 600, 8, 8, 8); dt = time.time() - t
    print(tag, "oracle 8x8 600^3:", round(dt, 3), "s, threads", O.num_threads(), "affinity", len(os.sched_getaffinity(0)), flush=True)
if order == "oracle_first":
    bench_oracle("before GPU init")
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print("affinity after GPU init", len(os.sched_getaffinity(0)))
bench_oracle("after GPU init")
