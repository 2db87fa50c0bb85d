import sys; sys.path.insert(0,'.')
import torch, os
MB=int(os.environ.get('SCDA_ADAM_BLOCKS','0'))
from scda_amd import native as N
n=136_850_000//4*4
dev=torch.device('cuda:0')
p=torch.randn(n,device=dev); g=torch.randn(n,device=dev)*1e-3; m=torch.zeros(n,device=dev); v=torch.zeros(n,device=dev)
def t(it=10):
    N.adam_step(p,g,m,v,1e-4,0.9,0.999,1e-8,1e-4,1,MB); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for i in range(it): N.adam_step(p,g,m,v,1e-4,0.9,0.999,1e-8,1e-4,i+2,MB)
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it
ms=t(); print("adam blocks=%d %.3f ms  %.2f TB/s" % (MB, ms, n*28/ms/1e9))
