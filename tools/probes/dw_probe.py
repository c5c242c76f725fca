import ctypes, os, sys, time
sys.path.insert(0, "deep-active-semantic-segmentation_amd")
import torch
from dass_hip._lib import lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (n,h,w,c,stride,dil) in [(16,129,129,144,1,1),(16,257,257,96,2,1),(16,33,33,960,1,2),(16,65,65,192,1,1)]:
    pad=dil; oh=(h+2*pad-2*dil-1)//stride+1; ow=(w+2*pad-2*dil-1)//stride+1
    x=torch.randn(n,h,w,c,device="cuda"); dy=torch.randn(n,oh,ow,c,device="cuda"); dw=torch.empty(c,9,device="cuda")
    for gy in ("2048","1024","512","256","4096"):
        os.environ["DASS_DW_GY"]=gy
        for it in range(6):
            if it==1: torch.cuda.synchronize(); t0=time.perf_counter()
            lib.dass_dwconv3x3_bwd_weight(P(x),c,P(dy),c,P(dw),n,h,w,c,oh,ow,stride,pad,dil,0,st)
        torch.cuda.synchronize(); us=(time.perf_counter()-t0)/5*1e6
        print((n,h,w,c,stride,dil),"gy",gy,"%.0f us %.2f TB/s"%(us,(x.numel()+dy.numel())*4/us/1e6))
