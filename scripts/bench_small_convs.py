import sys; sys.path.insert(0,"/root/repo")
import torch
from scda_amd import native as N
dev=torch.device("cuda:0")
def t(fn,it=50):
    fn(); torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it*1e3
for (B,Cin,H,W,Cout,k,s,p) in [(1,512,32,64,30,1,1,0),(1,512,32,64,60,1,1,0),(4,512,16,16,1,1,1,0),(4,256,16,16,1,1,1,0),(1,512,32,64,512,3,1,1),(4,3,256,256,64,3,2,1)]:
    x=torch.randn(B,Cin,H,W,device=dev); w=torch.randn(Cout,Cin,k,k,device=dev)*0.05
    y=N.conv2d_fwd(x,w,None,s,p,0); dy=torch.randn_like(y)
    print((B,Cin,H,W,Cout,k,s), "fwd %.1f us dgrad %.1f us wgrad %.1f us"%(t(lambda:N.conv2d_fwd(x,w,None,s,p,0)), t(lambda:N.conv2d_dgrad(dy,w,x.shape,s,p)), t(lambda:N.conv2d_wgrad(dy,x,w.shape,s,p))))
