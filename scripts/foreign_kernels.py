"""Which Python call sites run the ATen operators whose kernels are NOT this library's (at::native fills / copies / adds /
gathers, runtime copies)?  A TorchDispatchMode over ONE eager iteration (SCDA_GAN_GRAPH=0, so the GAN phases' operators are visible
too; the mode travels into the autograd engine's thread with the thread-local state) logs every operator that touches a device
tensor and can launch a kernel (views, metadata and allocation are skipped), with its tensor shapes and the innermost Python frame
under scda_amd/ (operators the engine runs by itself -- gradient accumulation, expand / sum backward -- have none and are listed
with the autograd node that issued them).

    python scripts/foreign_kernels.py [resnet50|maskrcnn] > gpurun_out/foreign_kernels.txt
"""
import sys, os, collections, traceback
os.environ.setdefault("SCDA_GAN_GRAPH", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten
from scda_amd.train_step import ScdaTrainer

which = sys.argv[1] if len(sys.argv) > 1 else "vgg16"
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
if which == "vgg16":
    tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
    src, tgt, gts, info = bench.synth_batch(0)
    masks = None
else:
    from scda_amd import resnet_config as RC
    tr = RC.make_trainer(bench.CFG, dev, lr=1.25e-5, world_size=1, with_mask=which == "maskrcnn",
                         mask_iou=0.2 if which == "maskrcnn" else None)
    src, tgt, gts, info = bench.synth_batch(0, RC.H, RC.W)
    masks = RC.synth_masks(gts, RC.H, RC.W) if which == "maskrcnn" else None
src, tgt = src.to(dev), tgt.to(dev)
kw = {} if masks is None else {"gt_masks": masks}
for i in range(6): tr.step(src, gts, info, tgt, **kw)
torch.cuda.synchronize()

NO_KERNEL = ("view", "reshape", "expand", "permute", "transpose", "select", "slice", "squeeze", "unsqueeze", "detach", "alias",
             "as_strided", "empty", "t.default", "size", "stride", "numel", "is_", "_unsafe_view", "unbind", "split", "narrow",
             "lift_fresh", "_local_scalar_dense", "resize_", "set_", "dim", "sym_", "record_stream", "_reshape_alias", "unfold")
log = collections.OrderedDict()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if any(name.startswith(p) for p in NO_KERNEL):
            return out
        flat, _ = tree_flatten((args, kwargs or {}))
        ts = [a for a in flat if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in ts):
            return out
        site = None
        for fr in reversed(traceback.extract_stack()[:-1]):
            if "scda_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                site = "%s:%d %s" % (fr.filename.split("scda_amd/")[-1], fr.lineno, fr.name)
                break
        shapes = " ".join("x".join(map(str, t.shape)) + ("" if t.is_contiguous() else "*") + ("" if t.is_cuda else "@cpu") for t in ts[:3])
        key = (site or "(engine)", name, shapes)
        rec = log.setdefault(key, [0, 0])
        rec[0] += 1
        rec[1] += sum(t.numel() * t.element_size() for t in ts[:3])
        return out


with Log():
    tr.step(src, gts, info, tgt, **kw)
torch.cuda.synchronize()
print("# ATen operators on device tensors in one eager iteration (%s): %d calls at %d distinct (site, op, shapes)" %
      (which, sum(r[0] for r in log.values()), len(log)))
print("%4s %10s  %-28s %-44s %s" % ("n", "bytes", "op", "shapes (* = strided)", "site"))
for (site, op, shapes), (n, b) in sorted(log.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    print("%4d %10d  %-28s %-44s %s" % (n, b, op[:28], shapes[:44], site))
