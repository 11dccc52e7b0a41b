"""Round 6: what plain PyTorch kernels stream on this box - the practical ceiling next to the 8 TB/s of the HBM roofline. One MI355X box:
bitwise_or(x, 1, out=y) 6.1-6.2 TB/s (read + write) at 512 MiB and 2 GiB, 7.2 at 128 MiB (both tensors fit the 256 MB Infinity Cache); the
long-K kernel streams the 128 MiB adjacency at 5.6-5.7 TB/s.  python tools/hbm_probe.py"""
import torch, time
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (128, 512, 2048):
    x = torch.randint(0, 2**31 - 1, (mb * 1024 * 1024 // 4,), dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    us = t(lambda: x.sum())
    print(f"{mb} MiB int32 sum: {us:.1f} us = {mb*1.048576e6/us/1e6:.2f} TB/s read")
    us = t(lambda: torch.bitwise_or(x, 1, out=y))
    print(f"{mb} MiB or->out: {us:.1f} us = {2*mb*1.048576e6/us/1e6:.2f} TB/s read+write")
    xf = x.view(torch.float32)
    us = t(lambda: xf.amax())
    print(f"{mb} MiB float amax: {us:.1f} us = {mb*1.048576e6/us/1e6:.2f} TB/s read")
