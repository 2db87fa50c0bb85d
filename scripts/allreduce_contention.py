"""What does a gradient all-reduce running BESIDE the iteration cost the kernels it overlaps?  One GPU cannot move bytes over xGMI, but
it can show the one multi-GPU cost that is local: a collective's kernels hold some CUs and stream the 547 MB detector bucket through
HBM while other kernels run.  Stand-in for RCCL's ring kernels (scripts/micro/ring_standin.hip): a PERSISTENT kernel of N workgroups
(RCCL runs one per channel, 16-32 channels) that stay resident and stream the bucket for as many passes as a ring all-reduce of 8
ranks moves (2 * 7/8 * 547 MB read + written ~ 4 passes), on a stream of its own.  Launch points: 'end' = behind the early detector
backward (one piece: round 4's), 'segmented' = the classifier + heads' 479 MB from the hook inside the backward, behind FC6's weight
gradient, + the conv body's 68 MB behind the backward (round 5: distributed_utils.SegmentedReduce).
    python scripts/allreduce_contention.py [workgroups=32] [passes=4] [heavy=0|1] [torch]
heavy = 1: the stand-in's workgroups cannot share a CU with a Winograd workgroup (64 KB of LDS; RCCL's ~128-register kernels cannot
either); `torch`: round 4's stand-in instead (element-wise torch kernels on a CU-masked stream: ~100 k short workgroups that starve
beside one-workgroup-per-CU launches and last for most of the iteration -- kept for comparison)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("SCDA_GAN_GRAPH", "0")
import numpy as np, torch, bench
from scda_amd import _timing as T
from scda_amd.train_step import ScdaTrainer

cus = int(sys.argv[1]) if len(sys.argv) > 1 else 32
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for _ in range(8):
    tr.step(src, gts, info, tgt)
torch.cuda.synchronize()

heavy = int(sys.argv[3]) if len(sys.argv) > 3 else 0
use_torch = len(sys.argv) > 4 and sys.argv[4] == "torch"
hip = ctypes.CDLL("libamdhip64.so")
mask = (ctypes.c_uint32 * 8)()
step = max(1, 256 // cus)
for i in range(0, 256, step):
    mask[i // 32] |= 1 << (i % 32)
raw = ctypes.c_void_p()
rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(raw), ctypes.c_uint32(8), mask)
assert rc == 0, rc
comm = torch.cuda.ExternalStream(raw.value, device=dev)
bucket = tr.flat['det'].grad
scratch = torch.empty_like(bucket)


flat = tr.flat['det']
span = flat.span_of([p for n_, p in tr.model.named_parameters() if n_.startswith(tuple(tr.model.EARLY_REDUCE_PREFIXES))])


ring = None if use_torch else ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "build", "libring_standin.so"))
if ring is not None:
    ring.standin_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    comm = torch.cuda.Stream(device=dev)       # a plain stream: the persistent kernel's grid IS the CU budget


def stand_in(lo, hi):
    comm.wait_stream(torch.cuda.current_stream())
    if ring is not None:
        lo, hi = lo // 4 * 4, hi // 4 * 4
        rc = ring.standin_launch(bucket.data_ptr() + 4 * lo, scratch.data_ptr() + 4 * lo, hi - lo, cus, passes, heavy, comm.cuda_stream)
        assert rc == 0, rc
        return
    with torch.cuda.stream(comm):
        for _ in range(passes):
            torch.add(bucket[lo:hi], 0.0, out=scratch[lo:hi])      # one read + one write of the slice


def measure(with_comm, n=12):
    """with_comm: False | 'end' (one stand-in behind the detector backward: round 4's launch point) | 'segmented' (the classifier +
    heads' slice from the hook INSIDE the backward, behind FC6's weight gradient, the conv body's behind the backward: round 5's)"""
    T.ENABLED = T.DEVICE = True
    acc, order, tot = {}, [], 0.0
    hook_orig, fwd_orig = tr._all_reduce, tr.model.forward

    def all_reduce(module, async_op):
        if with_comm and module is tr.model:
            stand_in(*((0, bucket.numel()) if with_comm == 'end' else (0, span[0])))
        return hook_orig(module, async_op)

    tr._all_reduce = all_reduce
    if with_comm == 'segmented':
        tr.model.forward = lambda x, t=None: fwd_orig(dict(x, _after_head_backward=lambda: stand_in(*span)), t)
    for _ in range(n):
        T.MARKS.clear(); T.EVENTS.clear()
        tr.step(src, gts, info, tgt)
        end = torch.cuda.Event(enable_timing=True); end.record()
        torch.cuda.current_stream().wait_stream(comm)
        torch.cuda.synchronize()
        ev = T.EVENTS + [("end", end)]
        for (la, a), (lb, b) in zip(ev, ev[1:]):
            acc[lb] = acc.get(lb, 0.0) + a.elapsed_time(b)
            if lb not in order: order.append(lb)
        tot += ev[0][1].elapsed_time(end)
    tr._all_reduce = hook_orig
    tr.model.forward = fwd_orig
    T.ENABLED = T.DEVICE = False
    return {k: v / n for k, v in acc.items()}, tot / n


base, tb = measure(False)
cont, tc = measure('end')
seg, ts = measure('segmented')
gan = ('crops+dec_fwd_enqueued', 'phase1', 'phase2', 'phase3')
print("stand-in (%s): %d passes over the %.0f MB detector bucket, %d workgroups / CUs; 'end' = launched behind the detector backward (one piece), "
      "'segmented' = %.0f MB from the hook behind FC6's weight gradient + %.0f MB behind the backward" % (
      "torch element-wise kernels on a CU-masked stream" if use_torch else "persistent ring_standin kernel%s" % (", exclusive CUs" if heavy else ""), passes, bucket.numel() * 4 / 1e6, cus,
      (span[1] - span[0]) * 4 / 1e6, span[0] * 4 / 1e6))
print("%-28s %9s %9s %9s" % ("segment (device ms)", "alone", "end", "segmented"))
for k in base:
    print("%-28s %9.2f %9.2f %9.2f" % (k, base[k], cont.get(k, float('nan')), seg.get(k, float('nan'))))
print("%-28s %9.2f %9.2f %9.2f" % ("decoder forward + phases 1-3", sum(base[k] for k in gan), sum(cont[k] for k in gan), sum(seg[k] for k in gan)))
print("%-28s %9.2f %9.2f %9.2f   (+%.1f %% / +%.1f %%)" % ("iteration", tb, tc, ts, 100.0 * (tc / tb - 1.0), 100.0 * (ts / tb - 1.0)))
