import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, QGTC
for (H,W,b,cm) in [(4096,4096,1,False),(8192,8192,1,False),(4096,4096,1,True),(8192,8192,1,True),(4096,4096,2,True)]:
    x=torch.rand(H,W,device='cuda')*2**b
    out=QGTC.val2bit(x,b,cm,False)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): o=QGTC.val2bit(x,b,cm,False)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/100
    print(os.environ.get("QGTC_PACK_EXP","default"), os.environ.get("QGTC_PACK_CCAP","-"), f"{H}x{W} b={b} cols={cm}: {us:.1f} us {(H*W*4+b*H*W/8)/us/1e6:.2f} TB/s")
