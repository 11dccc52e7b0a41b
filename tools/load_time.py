import time, sys, os
sys.path.insert(0, os.getcwd())
t0=time.time(); import torch; torch.cuda.init(); torch.zeros(1,device="cuda"); torch.cuda.synchronize(); t1=time.time()
import QGTC as Q; t2=time.time()
x=Q.val2bit(torch.rand(64,128,device="cuda"),2,False,False); torch.cuda.synchronize(); t3=time.time()
w=Q.val2bit(torch.rand(128,64,device="cuda"),2,True,False); o=Q.bitMM2Bit(x,w,64,128,64,2,2,2); torch.cuda.synchronize(); t4=time.time()
for i in range(3):
    t=time.time(); o=Q.bitMM2Bit(x,w,64,128,64,2,2,2); torch.cuda.synchronize(); print("call", round(time.time()-t,4))
print("torch init", round(t1-t0,2), "import QGTC", round(t2-t1,2), "first val2bit", round(t3-t2,2), "first mm", round(t4-t3,2))
import ctypes, qgtc_ppopp22_amd
t=time.time(); L=ctypes.CDLL(qgtc_ppopp22_amd.lib_path()); print("CDLL again", round(time.time()-t,3))
