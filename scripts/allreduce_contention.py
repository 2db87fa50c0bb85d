"""What does a gradient all-reduce running BESIDE the iteration cost the kernels it overlaps?  One GPU cannot move bytes over xGMI, but
it can show the one multi-GPU cost that is local: a collective's kernels hold some CUs and stream the 547 MB detector bucket through
HBM while the decoder / discriminator phases run.  Stand-in for RCCL's ring kernels: element-wise passes over the detector's gradient
bucket on a stream restricted to N CUs (hipExtStreamCreateWithCUMask; RCCL runs one workgroup per channel, 16-32 channels), started
behind the early detector backward exactly where ScdaTrainer launches the real all-reduce, for as many passes as a ring all-reduce of
8 ranks moves (2 * 7/8 * 547 MB read + written ~ 4 passes).  Reports the device time of the overlapped segment and of the iteration,
without / with the stand-in.   python scripts/allreduce_contention.py [cus=32] [passes=4]"""
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


def measure(with_comm, n=12):
    T.ENABLED = T.DEVICE = True
    acc, order, tot = {}, [], 0.0
    hook_orig = tr._all_reduce

    def all_reduce(module, async_op):
        if with_comm and module is tr.model:
            comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm):
                for _ in range(passes):
                    torch.add(bucket, 0.0, out=scratch)      # one read + one write of the 547 MB bucket
        return hook_orig(module, async_op)

    tr._all_reduce = all_reduce
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
    T.ENABLED = T.DEVICE = False
    return {k: v / n for k, v in acc.items()}, tot / n


base, tb = measure(False)
cont, tc = measure(True)
gan = ('crops+dec_fwd_enqueued', 'phase1', 'phase2', 'phase3')
print("stand-in: %d passes over the %.0f MB detector bucket on a %d-CU stream, launched where the detector's all-reduce is" % (passes, bucket.numel() * 4 / 1e6, cus))
print("%-28s %9s %9s" % ("segment (device ms)", "alone", "beside"))
for k in base:
    print("%-28s %9.2f %9.2f" % (k, base[k], cont.get(k, float('nan'))))
print("%-28s %9.2f %9.2f" % ("decoder forward + phases 1-3", sum(base[k] for k in gan), sum(cont[k] for k in gan)))
print("%-28s %9.2f %9.2f   (+%.1f %%)" % ("iteration", tb, tc, 100.0 * (tc / tb - 1.0)))
