import os, sys, traceback
os.environ["SCDA_GAN_GRAPH"]="1"; os.environ["SCDA_ALLOW_TEST_HOOKS"]="1"
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import numpy as np, torch
import model_common as mc
from test_train_step_gpu import build_product
from scda_amd import train_step as TS
cuda=torch.device('cuda:0')
orig=TS._GanGraphs.record
def rec(self,name,t,fn):
    print("recording", name, flush=True)
    try:
        r=orig(self,name,t,fn); print("  ok", name, flush=True); return r
    except Exception as e:
        print("  FAILED", name, type(e).__name__, str(e)[:200], flush=True); raise
TS._GanGraphs.record=rec
torch.manual_seed(1)
tr=TS.ScdaTrainer(mc.CFG,cuda,lr=1e-3,new_w=512,new_h=256,models=mc.seeded_models(build_product))
np.random.seed(5)
for it in range(4):
    src,tgt,gts,info=mc.seeded_inputs(256,512,sample=it%3)
    out=tr.step(src.to(cuda),gts,info,tgt.to(cuda)); print(it, float(out['loss']), flush=True)
