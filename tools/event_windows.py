import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, QGTC as Q
import bench
dev = torch.device("cuda:0")
M = K = 4096; N = 64; w = 1
A, X, bA, bX = bench.make_workload(Q, M, K, N, w, dev, seed=3)
out = Q.bitMM2Bit(bA, bX, M, K, N, 1, w, w)
enq = lambda n: Q.bitMM2Bit_enqueue(out, bA, bX, M, K, N, 1, w, w, n)
t = time.perf_counter()
while time.perf_counter() - t < 0.3:
    enq(200); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def win(n=200):
    e0.record(); enq(n); e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 3)
print("windows before:", [win() for _ in range(5)])
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); enq(1000); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("1000-launch region wall us/step", round((t1 - t0) * 1e3, 3))
    print("windows after:", [win() for _ in range(8)])
print("window of 1000:", win(1000), win(1000))
print("windows after:", [win() for _ in range(8)])
time.sleep(0.01)
print("after 10 ms idle:", [win() for _ in range(4)])
print("--- bench.time_steps itself ---")
for steps in (1000, 1000, 200, 1000):
    wall, kern, wall_ev, _, _ = bench.time_steps(Q, out, bA, bX, M, K, N, w, steps, 50, lambda: None)
    print(steps, "wall us/step", round(wall / steps * 1e6, 3), "event us/launch", round(kern * 1e6, 3), "event-region wall us/step", round(wall_ev / steps * 1e6, 3))
