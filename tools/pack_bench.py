import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC, time
for (H,W,b,cm) in [(4096,4096,1,False),(4096,4096,2,False),(8192,8192,1,False),(1216,1213,1,False),(1216,1216,1,False),(4096,64,1,True),(4096,64,8,True),(1213,128,2,True),(4096,4096,1,True),(4096,4096,2,True),(8192,8192,1,True)]:
    x=torch.rand(H,W,device='cuda')*2**b
    for _ in range(3): o=QGTC.val2bit(x,b,cm,False)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): o=QGTC.val2bit(x,b,cm,False)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/50
    byts=H*W*4+b*H*W/8
    print(f"val2bit {H}x{W} b={b} col_major={cm}: {us:.1f} us  {byts/us/1e6:.2f} TB/s algorithmic")
