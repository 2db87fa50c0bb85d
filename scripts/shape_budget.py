"""Where the MFMA time of one iteration goes, by (operation, shape): every conv2d_fwd / dgrad / wgrad / gemm call of one
ScdaTrainer.step is recorded (monkey-patched scda_amd.native), then each distinct call is replayed in isolation and timed.
Output: launches/iteration, isolated us, GFLOP, TFLOP/s, share of the isolated-sum budget -- sorted by time.
    gpurun -- 'python scripts/shape_budget.py > gpurun_out/shape_budget.txt'"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from scda_amd import native
from scda_amd.train_step import ScdaTrainer

dev = torch.device("cuda:0")
torch.manual_seed(0)
np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0)
src, tgt = src.to(dev), tgt.to(dev)
for _ in range(4):
    tr.step(src, gts, info, tgt)
torch.cuda.synchronize()

calls = {}


def rec(key, flops, replay):
    e = calls.setdefault(key, [0, flops, replay])
    e[0] += 1


orig = {k: getattr(native, k) for k in ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad", "conv2d_wgrad_bias", "gemm")}


def conv_flops(xs, ws, ys):
    return 2.0 * ws[0] * ws[1] * ws[2] * ws[3] * ys[0] * ys[2] * ys[3]


def p_fwd(x, w, bias, stride, pad, act=0, slope=0.01, row_period=0):
    y = orig["conv2d_fwd"](x, w, bias, stride, pad, act, slope, row_period=row_period)
    xs, ws = tuple(x.shape), tuple(w.shape)
    rec(("fwd", xs, ws, stride, pad), conv_flops(xs, ws, y.shape),
        lambda a=torch.randn(xs, device=dev), b=torch.randn(ws, device=dev) * .05: orig["conv2d_fwd"](a, b, None, stride, pad, act, slope))
    return y


def p_dgrad(dy, w, x_shape, stride, pad, act_src=None, act_slope=0.0, row_period=0):
    r = orig["conv2d_dgrad"](dy, w, x_shape, stride, pad, act_src, act_slope, row_period=row_period)
    ys, ws, xs = tuple(dy.shape), tuple(w.shape), tuple(x_shape)
    rec(("dgrad", xs, ws, stride, pad), conv_flops(xs, ws, ys),
        lambda a=torch.randn(ys, device=dev), b=torch.randn(ws, device=dev) * .05: orig["conv2d_dgrad"](a, b, xs, stride, pad))
    return r


def p_wgrad(dy, x, w_shape, stride, pad, out=None, **kw):
    r = orig["conv2d_wgrad"](dy, x, w_shape, stride, pad, out, **kw)
    ys, ws, xs = tuple(dy.shape), tuple(w_shape), tuple(x.shape)
    rec(("wgrad", xs, ws, stride, pad), conv_flops(xs, ws, ys),
        lambda a=torch.randn(ys, device=dev), b=torch.randn(xs, device=dev): orig["conv2d_wgrad"](a, b, ws, stride, pad))
    return r


def p_wgrad_bias(dy, x, w_shape, stride, pad, out=None, db_out=None, **kw):
    r = orig["conv2d_wgrad_bias"](dy, x, w_shape, stride, pad, out, db_out, **kw)
    ys, ws, xs = tuple(dy.shape), tuple(w_shape), tuple(x.shape)
    rec(("wgrad+b", xs, ws, stride, pad), conv_flops(xs, ws, ys),
        lambda a=torch.randn(ys, device=dev), b=torch.randn(xs, device=dev): orig["conv2d_wgrad_bias"](a, b, ws, stride, pad))
    return r


def p_gemm(a, b, M, N, K, lda, ldb, trans_a=False, trans_b=False, *args, **kw):
    r = orig["gemm"](a, b, M, N, K, lda, ldb, trans_a, trans_b, *args, **kw)
    sa, sb = tuple(a.shape), tuple(b.shape)
    rec(("gemm", (M, N, K), (trans_a, trans_b), 0, 0), 2.0 * M * N * K,
        lambda a=torch.randn(sa, device=dev), b=torch.randn(sb, device=dev): orig["gemm"](a, b, M, N, K, lda, ldb, trans_a, trans_b))
    return r


native.conv2d_fwd, native.conv2d_dgrad, native.conv2d_wgrad, native.conv2d_wgrad_bias, native.gemm = p_fwd, p_dgrad, p_wgrad, p_wgrad_bias, p_gemm
import scda_amd.autograd_ops as AO  # noqa: E402  (modules that did `from .native import x` keep the originals: patch both)
for k, f in (("conv2d_fwd", p_fwd), ("conv2d_dgrad", p_dgrad), ("conv2d_wgrad", p_wgrad), ("conv2d_wgrad_bias", p_wgrad_bias), ("gemm", p_gemm)):
    if hasattr(AO, k):
        setattr(AO, k, f)
tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
for k, f in orig.items():
    setattr(native, k, f)


def timeit(fn, it=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


rows = []
for key, (n, fl, replay) in calls.items():
    try:
        us = timeit(replay)
    except Exception as e:  # noqa: BLE001
        print("replay failed", key, e)
        continue
    rows.append((n * us, n, us, fl, key, native.last_plan() if hasattr(native, "last_plan") else None))
tot = sum(r[0] for r in rows)
totf = sum(r[1] * r[3] for r in rows)
print("isolated-sum budget: %.2f ms / iteration, %.3f TFLOP -> %.1f TFLOP/s" % (tot / 1e3, totf / 1e12, totf / tot / 1e6))
print("%6s %3s %9s %8s %7s %6s  %s" % ("share", "n", "us each", "GFLOP", "TF/s", "cum%", "op shape"))
cum = 0
for t, n, us, fl, key, plan in sorted(rows, key=lambda r: -r[0]):
    cum += t
    print("%5.1f%% %3d %9.1f %8.2f %7.1f %5.1f%%  %s %s" % (100 * t / tot, n, us, fl / 1e9, fl / us / 1e6, 100 * cum / tot, key, plan))
