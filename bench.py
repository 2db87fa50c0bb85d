#!/usr/bin/env python
"""bench.py -- the SCDA hot path on MI355X: VGG16 Faster-R-CNN + 4-cluster SCDA training iteration
(forward + backward + gradient all-reduce + Adam, all four optimiser phases), synthetic 512x1024 input, batch 1/GPU.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either started by itself -- it then launches N ranks, one per GPU -- or under
   `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one iteration = 1 source + 1 target 512x1024 image per GPU; images/s = 2 * N * steps / time (as in the
reference, the source image goes forward AND backward, the target image forward only -- it has no loss; `config.iters_per_s`
is the iteration rate for comparisons with iteration-based numbers).
Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (fp32-MFMA implicit-GEMM conv forward with direct-to-LDS staging, 3x3 stride 1, 128- or 256-row tiles): algorithmic FLOPs of its
                launches inside the timed region / their summed duration (HIP events recorded on the launch stream by the
                library's profiler), against the 157.3 TFLOP/s fp32-MFMA peak
                `roofline.iteration` = the whole iteration against the same peak: F_iter (2.255 TFLOP of necessary conv / FC
                work per iteration, SURVEY.md 8d / BASELINE.md 2) x iterations/s
  cpu_baseline  the CPU oracle of the same iteration (oracle/torch_ref.py, a faithful PyTorch-CPU restatement of the
                reference's train() body, pinned against the reference) timed on this box's host cores: rank 0, N=1 only;
                1 warm-up + 3 timed iterations
Before timing, the detector is pre-conditioned: warm-up iterations continue (up to 40) until the RPN yields the
full post-NMS quota on both images, so that NMS / RoI sampling see realistic proposal sets; the counts are in the JSON.
"""
import argparse
import json
import os
import sys
import time

# the host driver of the MI355X boxes only supports dmabuf IPC: without this RCCL's intra-node set-up (and any CUDA-tensor sharing
# across processes) fails with hipIpcGetMemHandle: invalid argument.  Must be in the environment before HIP initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _early_world():
    """ranks of this run as far as the command line / environment say, BEFORE torch is imported: the data-parallel process
    environment (scda_amd.hostenv.data_parallel_env: eight hardware queues ...) must be in place when HIP initialises, and
    launch_ranks' children inherit it"""
    n = int(os.environ.get("WORLD_SIZE", "1") or 1)
    for i, tok in enumerate(sys.argv):
        if tok == "--gpus" and i + 1 < len(sys.argv) and sys.argv[i + 1].isdigit():
            n = max(n, int(sys.argv[i + 1]))
        elif tok.startswith("--gpus=") and tok[7:].isdigit():
            n = max(n, int(tok[7:]))
    return n


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scda_amd import hostenv as _hostenv                       # (imports nothing of torch)
_BLOCKING = _hostenv.wants_blocking_sync(_early_world())
_DP_ENV = _hostenv.data_parallel_env(_early_world())

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = {
    "shared": {"gan_model_flag": 2, "scales": [512], "max_size": 1024, "anchor_scales": [2, 4, 8, 16, 32],
               "anchor_ratios": [0.5, 1, 2], "anchor_stride": 16, "bbox_normalize_stats_precomputed": True,
               "bbox_normalize_stds": [0.1, 0.1, 0.2, 0.2], "bbox_normalize_means": [0, 0, 0, 0], "num_classes": 9},
    "train_anchor_target_cfg": {"rpn_batch_size": 256, "nms_iou_thresh": 0.7, "positive_iou_thresh": 0.7,
                                "negative_iou_thresh": 0.3, "positive_percent": 0.5, "ignore_iou_thresh": 0.5},
    "train_rpn_proposal_cfg": {"nms_iou_thresh": 0.7, "pre_nms_top_n": 12000, "post_nms_top_n": 2000, "roi_min_size": 2},
    "train_proposal_target_cfg": {"batch_size": 512, "positive_iou_thresh": 0.5, "negative_iou_thresh_hi": 0.5,
                                  "negative_iou_thresh_lo": 0.0, "ignore_iou_thresh": 0.5, "positive_percent": 0.25,
                                  "append_gts": True},
    "test_rpn_proposal_cfg": {"nms_iou_thresh": 0.7, "pre_nms_top_n": 6000, "post_nms_top_n": 300, "roi_min_size": 2},
    "test_predict_bbox_cfg": {"nms_iou_thresh": 0.5, "score_thresh": 0.0, "top_n": 100},
}
for _k in CFG:
    if _k != "shared":
        CFG[_k].update(CFG["shared"])

H, W, G = 512, 1024, 12
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
F_ITER_TFLOP = 2.255          # necessary conv / FC / convT work of one iteration
PMC_TRAFFIC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
# Winograd-eligible share of F_iter (direct-form TFLOP of the launches the F(2x2,3x3) kernels take; scripts/wino_iteration_flops.py
# sums the library's launch log of one iteration): detector forward + data gradient 0.9857, its weight gradients 0.3286, the SCDA
# nets' stride-1 3x3 layers (decoder residual / up-sampling convolutions, forward + both gradients).  F_exec = F_iter - eligible * (1 - 1/2.25)
WINO_ELIGIBLE_TFLOP = {"detector fwd + dgrad": 0.9857, "detector wgrad": 0.3286, "scda nets": 0.3600}
DOMINANT_SMALL_TILES = "conv_igemm_glds_kernel<64,*,3,3,1,fwd>"
DOMINANT = "conv_igemm_glds_kernel<128|256,*,3,3,1,fwd>"   # the instantiations with 128 or 256 tile rows (256 = 8 waves), any tile width, 3x3 stride 1, forward
# with the Winograd kernel (the default): every stride-1 3x3 forward with >= 64 channels is a launch of this ONE kernel (VGG 12 + RPN 1
# per image, the decoders' residual / up-sampling convolutions); the library's profiler counts the MFMA work it EXECUTES
DOMINANT_WINO = ("conv_wino_kernel<fwd>", "conv_wino_kernel<dgrad>")   # ONE kernel function (conv_wino_kernel<1|2> in rocprofv3's listing): the
#                                       data gradient is the same launch with the rotated filters; the profiler tags them apart
WINO_RATIO = 2.25            # direct-convolution MACs per executed Winograd F(2x2,3x3) MAC


def cpu_baseline_resnet(with_mask):
    """BASELINE configs[3] / [4] on the host cores: the DETECTOR part of the iteration through the CPU oracle of that configuration
    (oracle/resnet_ref.py: source forward with the four / five losses, target forward, backward) at 800 x 1344.  No CPU statement of
    the SCDA nets exists on this configuration's rectangular maps inside one trainer, so the figure omits them: an UPPER bound of the
    CPU rate (the SCDA nets are ~20 % of the VGG iteration's CPU time)."""
    from oracle import resnet_ref as RR, torch_ref as R
    from scda_amd import resnet_config as RC
    from scda_amd.hostenv import cpu_quota
    cores = os.cpu_count() or 1
    quota = cpu_quota()
    if quota:
        cores = min(cores, quota)
    torch.set_num_threads(cores)
    R.use_cpu_backend()
    try:
        torch.manual_seed(0)
        np.random.seed(0)
        shared = dict(CFG["shared"], with_mask=with_mask)
        det = RR.RefResNetDetector(shared).train()
        src, tgt, gts, info = synth_batch(0, RC.H, RC.W)
        x = {"cfg": CFG, "image": src, "image_info": info, "ground_truth_bboxes": gts, "ignore_regions": None, "cluster_num": 4,
             "threshold": 128}
        if with_mask:
            x["ground_truth_masks"] = RC.synth_masks(gts, RC.H, RC.W)
        times = []
        for i in range(3):
            t0 = time.time()
            out = det(x, tgt)
            sum(out["losses"]).backward()
            det.zero_grad()
            if i:
                times.append(time.time() - t0)
        dt = sum(times) / len(times)
    finally:
        R.reset_backend()
        torch.set_num_threads(1)
    return {"value": round(2.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "2 timed detector passes after 1 warm-up (source forward + losses, target forward, backward; 800x1344) of "
                      "oracle/resnet_ref.py RefResNetDetector%s -- the SCDA nets are not in it (upper bound of the CPU rate): %s s"
                      % (" with the mask branch" if with_mask else "", ", ".join("%.2f" % t for t in times)), "s_per_iter": round(dt, 3)}


def synth_batch(rank, H=H, W=W):
    """N(0,1) images clamped to [-1,1]; G integer-cornered gt boxes, log-uniform sizes, classes 1..8"""
    g = torch.Generator().manual_seed(1000 + rank)
    src = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1)
    tgt = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1)
    r = np.random.RandomState(2000 + rank)
    w = np.exp(r.uniform(np.log(16), np.log(400), G)); h = np.exp(r.uniform(np.log(16), np.log(300), G))
    x1 = r.uniform(0, W - 1 - w); y1 = r.uniform(0, H - 1 - h)
    box = np.stack([np.floor(x1), np.floor(y1), np.minimum(np.ceil(x1 + w), W - 1), np.minimum(np.ceil(y1 + h), H - 1)], 1)
    gts = torch.from_numpy(np.concatenate([box, r.randint(1, 9, (G, 1))], 1).astype(np.float32)[None])
    return src, tgt, gts, torch.tensor([[H, W, 1.0]])


def pmc_traffic(kernel, prefix=""):
    """-> (bytes per launch, source file, launches averaged, algorithmic bytes per launch of the SAME launches or None, traffic / 8-L2 floor or None)"""
    t = _pmc_traffic(kernel, prefix)
    return t if t[0] is not None else (None, None, None, None, None)


def _pmc_traffic(kernel, prefix=""):
    """L2 memory-side bytes per launch of the dominant kernel.  PMC counters cannot be sampled from inside the timed
    run; they come from the committed rocprofv3 passes of this same command (scripts/collect_profiles.sh ->
    profiles/r01_pmc_traffic.json: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  None when that file is absent."""
    for name in PMC_TRAFFIC_FILES:     # newest committed counter pass first
        try:
            name = name.replace("_pmc_traffic", "_" + prefix + "pmc_traffic") if prefix else name
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if d["dominant"].get("kernel", DOMINANT) != kernel:
                continue              # a counter pass of another kernel (or another class of it) says nothing about this one
            like = d.get("dominant_detector_launches")
            if like:                  # joined per dispatch: exactly the launches the in-library profiler times (detector forward + data gradient)
                return (int(like["traffic_bytes_per_launch"]), "profiles/" + name, int(like["launches"]), int(like["algorithmic_bytes_per_launch"]),
                        like.get("traffic_over_xcd_floor"))
            return int(d["dominant"]["traffic_bytes_per_launch"]), "profiles/" + name, int(d["dominant"].get("launches", 0)), None, None
        except Exception:
            continue
    return None, None, None, None, None


def cpu_baseline():
    """one full-size iteration of the CPU oracle on this box's host cores (bounded sample of the same workload)"""
    from oracle import torch_ref as R
    from scda_amd.hostenv import cpu_quota
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        cores = os.cpu_count() or 1
    quota = cpu_quota()           # the container's CFS quota is what the host actually grants (16 CPUs on the bench boxes)
    if quota:
        cores = min(cores, quota)
    torch.set_num_threads(cores)
    R.use_cpu_backend()
    try:
        torch.manual_seed(0)
        np.random.seed(0)
        models = R.build_models(CFG)
        tr = R.RefTrainer(CFG, models, lr=1.25e-5, new_w=W, new_h=H)
        src, tgt, gts, info = synth_batch(0)
        tr.step(src, gts, info, tgt)            # warm-up (allocator, oneDNN primitive caches)
        times = []
        for _ in range(3):
            t0 = time.time()
            tr.step(src, gts, info, tgt)
            times.append(time.time() - t0)
        dt = sum(times) / len(times)
    finally:
        R.reset_backend()
    return {"value": round(2.0 / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "3 timed full iterations after 1 warm-up (each: 1 source + 1 target 512x1024 image, all 4 phases incl. "
                      "Adam) of oracle/torch_ref.py RefTrainer: %s s" % ", ".join("%.2f" % t for t in times),
            "s_per_iter": round(dt, 3)}


def launch_ranks(n):
    """`python bench.py --gpus N` started by itself (no torchrun around it): start N copies of this command, one process per GPU,
    as the reference's launcher does (examples/faster-rcnn/cityscapes/vgg/4cluster.sh:13-38: one task per GPU), with the env://
    rendezvous variables torch.distributed.run would set.  Children inherit stdout / stderr: rank 0 prints the one JSON line.
    -> exit code (0 iff every rank returned 0); a failing rank takes the others down (they would hang in a collective)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live = 0, list(procs)
    try:
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:          # exactly the processes started above, by handle
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)    # ~4 s of device work: long enough for an outside observer (the driver's gpu_busy sampler) to see the GPU
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", choices=["vgg16", "resnet50", "maskrcnn"], default="vgg16",
                    help="vgg16 = BASELINE.json configs[1] (the metric's configuration, default); resnet50 = configs[3]'s detector "
                         "(ResNet-50 C4 + SCDA at 800x1344, performance-only: the reference has no runnable model for it)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # test hooks (tests/test_bench_multirank_gpu.py runs the N > 1 branch with two ranks on ONE GPU over gloo)
    backend = os.environ.get("SCDA_BENCH_BACKEND", "nccl")
    if "SCDA_BENCH_DEVICE" in os.environ:
        local = int(os.environ["SCDA_BENCH_DEVICE"])
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))   # plain `python bench.py --gpus N`: become the launcher, one rank per GPU
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start `python bench.py --gpus %d` by itself, or under "
                         "`python -m torch.distributed.run --nproc-per-node %d`" % (a.gpus, world, a.gpus, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if _BLOCKING:
        from scda_amd.hostenv import prefer_blocking_sync
        prefer_blocking_sync(local)  # ranks share few host cores (CPU quota < 2.25 x ranks): sleep, do not spin, while waiting for the device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        # the line printed below says n_gpus = world: refuse to print it for a group of any other size
        assert dist.get_world_size() == a.gpus == world, (dist.get_world_size(), a.gpus, world)

    from scda_amd import native
    from scda_amd.train_step import ScdaTrainer
    from scda_amd.dropin.utils.distributed_utils import broadcast_params

    torch.manual_seed(0)          # identical initial weights on every rank (then broadcast, as the reference does)
    np.random.seed(100 + rank)    # per-rank sampling / soft-label stream
    wino = native.wino_enabled()
    bh, bw, f_iter, dominant = H, W, F_ITER_TFLOP, (DOMINANT_WINO if wino else DOMINANT)
    step_kw = {}
    if a.config in ("resnet50", "maskrcnn"):
        from scda_amd import resnet_config as RC
        bh, bw, f_iter, dominant = RC.H, RC.W, RC.f_iter_tflop(), RC.DOMINANT
        # configs[4]: + mask branch and mask loss.  The untrained RPN of a synthetic run puts almost no proposal on an object at
        # IoU > 0.5, so the branch would see the 12 appended ground-truth boxes only: IoU > 0.2 fills its quota of 64 RoIs (the work
        # a trained model does); F_iter below counts the RoIs the branch actually ran on.
        tr = RC.make_trainer(CFG, dev, lr=1.25e-5, world_size=world, with_mask=a.config == "maskrcnn",
                             mask_iou=0.2 if a.config == "maskrcnn" else None)
    else:
        tr = ScdaTrainer(CFG, dev, lr=1.25e-5, new_w=W, new_h=H, world_size=world)
    if world > 1:
        for m in (tr.model, tr.dis, tr.dec, tr.dis_patch):
            broadcast_params(m)
    src, tgt, gts, info = synth_batch(rank, bh, bw)
    src, tgt = src.to(dev), tgt.to(dev)
    if a.config == "maskrcnn":
        step_kw["gt_masks"] = RC.synth_masks(gts, bh, bw)

    quota = CFG["train_rpn_proposal_cfg"]["post_nms_top_n"]
    precond = 0
    for _ in range(a.warmup):
        tr.step(src, gts, info, tgt, **step_kw)
    def short_of_quota():
        """any rank still below the post-NMS quota?  (decided collectively: every step contains all-reduces, so all ranks must
        run the same number of pre-conditioning steps)"""
        flag = torch.tensor([1 if min(tr.last_num_proposals) < quota else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return bool(flag.item())

    while precond < 40 and short_of_quota():   # realistic proposal sets before timing
        tr.step(src, gts, info, tgt, **step_kw)
        precond += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # event pairs around the dominant kernel only: they are queue markers.  VGG (the judged configuration): 24 launches per
    # iteration, timed INSIDE the timed region.  ResNet: its dominant class is 86 launches per iteration, so it is timed in a
    # separate pass after the timed region (3 more iterations) and the throughput number stays unperturbed.
    in_region = a.config == "vgg16"
    dom_list = list(dominant) if isinstance(dominant, tuple) else [dominant]
    # An event pair is two queue markers, and a launch behind a marker starts ~6 us after the kernel in front of it has ended where
    # an unmarked one follows back to back: with the pairs around all 39 Winograd launches of EVERY timed iteration the iteration is
    # 0.31 ms (1.7 %) longer than without them (scripts/event_pair_cost.py: alternating blocks in one process, 18.44 vs 18.13 ms).
    # They are therefore recorded in every PROF_EVERY-th timed iteration -- still inside the timed region, 39 x steps / PROF_EVERY
    # samples; SCDA_BENCH_PROF_EVERY=1 marks every iteration, as rounds 1 - 4 did.
    prof_every = max(1, int(os.environ.get("SCDA_BENCH_PROF_EVERY", "5")))
    prof_iters = 0
    # per-segment device time: in the SAME marked iterations an event goes onto the compute stream at each of the trainer's phase
    # marks (scda_amd/_timing.py; ~12 queue markers per marked iteration, no profiler), read back after the timed region.
    # host: CPU time of the main thread (it enqueues everything; it spins while it waits for a device result) and of the whole
    # process, over the timed region.
    from scda_amd import _timing as T
    seg_events = []
    torch.cuda.synchronize()
    cpu_main0, cpu_proc0 = time.thread_time(), time.process_time()
    t0 = time.perf_counter()
    for i in range(a.steps):
        marked = in_region and i % prof_every == min(2, prof_every - 1, a.steps - 1)     # (not the first iterations behind the barrier)
        if marked:
            native.prof_enable(dom_list)
            prof_iters += 1
            T.MARKS.clear(); T.EVENTS.clear()
            T.ENABLED = T.DEVICE = True
        out = tr.step(src, gts, info, tgt, **step_kw)
        if marked:
            native.prof_enable(False)
            T.ENABLED = T.DEVICE = False
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            seg_events.append(list(T.EVENTS) + [("end", end)])
    torch.cuda.synchronize()
    cpu_main, cpu_proc = time.thread_time() - cpu_main0, time.process_time() - cpu_proc0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    in_region_timed = in_region
    mask_rois = getattr(tr.model, "last_mask_rois", None) if a.config == "maskrcnn" else None
    if mask_rois is not None:
        f_iter += RC.mask_branch_tflop(mask_rois)
    if not in_region:
        native.prof_enable(dom_list)
        for _ in range(3):
            tr.step(src, gts, info, tgt, **step_kw)
        torch.cuda.synchronize()
    native.prof_enable(False)
    prof = native.prof_collect()
    template = None
    if os.environ.get("SCDA_BENCH_NO_TEMPLATE_PASS"):     # profile collection: keep the trace to the warm-up + timed iterations
        in_region = False
        template = None
    if wino:
        in_region = False     # one kernel, one tile shape: there is no "template over every tile shape" to add
    if in_region and rank == 0:
        # the same kernel TEMPLATE over every tile shape it is launched with (the 64-row instantiations of the 64-channel layers and
        # of the GAN nets included): ~3x the launches, so its event pairs go into a pass of their own after the timed region
        native.prof_enable([DOMINANT, DOMINANT_SMALL_TILES])
    if in_region:
        for _ in range(3):
            tr.step(src, gts, info, tgt, **step_kw)
        torch.cuda.synchronize()
        native.prof_enable(False)
        p2 = native.prof_collect()
        if rank == 0 and DOMINANT in p2:
            n2 = sum(p2[k][0] for k in (DOMINANT, DOMINANT_SMALL_TILES) if k in p2)
            ms2 = sum(p2[k][1] for k in (DOMINANT, DOMINANT_SMALL_TILES) if k in p2)
            fl2 = sum(p2[k][2] for k in (DOMINANT, DOMINANT_SMALL_TILES) if k in p2)
            template = {"kernel": "conv_igemm_glds_kernel<*,*,3,3,1,fwd> (every tile shape)", "launches_per_iteration": n2 // 3,
                        "gflop_per_iteration": round(fl2 / 3 / 1e9, 1), "ms_per_iteration": round(ms2 / 3, 3),
                        "achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 2), "frac": round(fl2 / (ms2 * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                        "how": "3 extra iterations after the timed region, HIP events on the launch stream"}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    segments = None
    if seg_events:
        acc, order = {}, []
        for ev in seg_events:
            for (la, ea), (lb, eb) in zip(ev, ev[1:]):
                acc[lb] = acc.get(lb, 0.0) + ea.elapsed_time(eb)
                if lb not in order:
                    order.append(lb)
        n_it = len(seg_events)
        segments = {"unit": "ms of device time on the compute stream, segment ENDING at the named mark (scda_amd/train_step.py); "
                            "average of %d marked iterations inside the timed region" % n_it}
        segments.update({k: round(acc[k] / n_it, 3) for k in order})
        segments["step_begin->end"] = round(sum(ev[0][1].elapsed_time(ev[-1][1]) for ev in seg_events) / n_it, 3)
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = 2.0 * world * a.steps / dt
        roof = None
        if dominant is None and prof:
            dominant = max(prof, key=lambda k: prof[k][1])
        if isinstance(dominant, tuple) and any(k in prof for k in dominant):     # classes of one kernel function: summed
            prof["conv_wino_kernel<1|2> (forward + data gradient)"] = tuple(sum(prof[k][i] for k in dominant if k in prof) for i in range(4))
            dominant = "conv_wino_kernel<1|2> (forward + data gradient)"
        if dominant in prof:
            n, tms, fl, by = prof[dominant]
            ach = fl / (tms * 1e-3) / 1e12
            # the committed counter passes of THIS configuration (none exist for the mask configuration: traffic stays null there)
            traffic, src, t_n, t_alg, t_floor = pmc_traffic(dominant, {"vgg16": "", "resnet50": "resnet50_", "maskrcnn": "maskrcnn_"}[a.config])
            it_ach = f_iter * world * a.steps / dt
            f_exec = f_iter - sum(WINO_ELIGIBLE_TFLOP.values()) * (1.0 - 1.0 / WINO_RATIO) if (wino and a.config == "vgg16") else None
            roof = {"bound": "mfma", "kernel": dominant, "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "bytes/launch (HBM+Infinity-Cache side of L2: 2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": src,
                    "algorithmic_bytes_per_launch": round(by / n), "launches": n, "avg_launch_ms": round(tms / n, 4),
                    "how": ("HIP event pairs on the launch stream around every launch of the class in %d of the %d timed iterations "
                            "(every %d%s: a pair costs the launch behind it ~6 us, scripts/event_pair_cost.py)"
                            % (prof_iters, a.steps, prof_every, "th" if prof_every > 1 else "")) if in_region_timed else
                           "HIP event pairs on the launch stream in 3 extra iterations after the timed region",
                    # like for like: the counter passes' average over the same launch population (joined per dispatch with the layer
                    # each ran), against THAT population's algorithmic bytes
                    "traffic_launches": t_n, "traffic_algorithmic_bytes_per_launch": t_alg,
                    "traffic_over_algorithmic": round(traffic / t_alg, 3) if traffic and t_alg else None,
                    # ... and against what eight non-coherent per-XCD L2s must fetch at least for the same launches (scripts/pmc_summary.py
                    # xcd_floor: (8 / gm) x transformed filters + gm x input): the counter sits at the L2s' memory side
                    "traffic_over_xcd_floor": t_floor,
                    "gflop_per_launch": round(fl / n / 1e9, 2),
                    **({"flop_definition": "MFMA work the Winograd F(2x2,3x3) kernel executes (16 products per 2x2 output tile and channel "
                                           "pair) = the direct convolution's / 2.25",
                        "direct_equivalent": {"achieved": round(ach * WINO_RATIO, 2), "frac": round(ach * WINO_RATIO / PEAK_F32_MFMA_TFLOPS, 4),
                                              "gflop_per_launch": round(fl * WINO_RATIO / n / 1e9, 2),
                                              "definition": "2 * Cout * pixels * Cin * 9 FLOP of the same launches / the same time (what "
                                                            "roofline.iteration's F_iter counts): may exceed the MFMA peak"}} if wino else {}),
                    "template_all_tiles": template,
                    "iteration": {"achieved": round(it_ach / world, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s per GPU",
                                  "frac": round(it_ach / world / PEAK_F32_MFMA_TFLOPS, 4),
                                  "definition": "F_iter (%.3f TFLOP of necessary conv/FC work per iteration) x iterations/s" % f_iter},
                    "iteration_executed": None if f_exec is None else {
                        "achieved": round(f_exec * a.steps / dt, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s per GPU",
                        "frac": round(f_exec * a.steps / dt / PEAK_F32_MFMA_TFLOPS, 4), "f_exec_tflop": round(f_exec, 3),
                        "definition": "F_exec = F_iter with the Winograd-eligible %.3f TFLOP counted at the 16 / 36 of them the kernels "
                                      "execute; x iterations/s -- a fraction of a bound again" % sum(WINO_ELIGIBLE_TFLOP.values())}}
        res = {
            "metric": "images/sec (fwd+bwd) VGG16 Faster-RCNN+SCDA 512x1024" if a.config == "vgg16" else
                      "images/sec (fwd+bwd) ResNet-50-C4 Faster-RCNN+SCDA 800x1344 (performance-only configuration)"
                      if a.config == "resnet50" else
                      "images/sec (fwd+bwd) ResNet-50-C4 Mask-RCNN+SCDA 800x1344 (performance-only configuration)",
            "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "collective": {"backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if world > 1 else None,
                           "ranks": dist.get_world_size() if world > 1 else 1,
                           # dis, dis_patch, dec + the detector's bucket in two pieces (classifier + heads from inside the backward)
                           "all_reduces_per_step": (5 if getattr(tr, "_det_early_span", lambda: None)() else 4) if world > 1 else 0,
                           # process environment of a rank (scda_amd.hostenv.data_parallel_env) and how its host thread waits
                           "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                           "host_wait": "blocking" if _BLOCKING else "spin", "env_set": _DP_ENV},
            "config": {"workload": ("vgg16_FasterRCNN + 4-cluster SCDA, synthetic Cityscapes->Foggy 512x1024, batch=1/GPU "
                                    "(BASELINE.json configs[1]); 1 step = 1 source + 1 target image, 4 optimiser phases")
                       if a.config == "vgg16" else
                       ("resnet50_FasterRCNN (C4, RoIAlignAvg, layer4 head) + 4-cluster SCDA, synthetic 800x1344, batch=1/GPU "
                        "(BASELINE.json configs[3]'s detector; no reference implementation exists: performance only)")
                       if a.config == "resnet50" else
                       ("mask_rcnn: the configs[3] detector + mask branch (RoIAlignAvg 14x14, 4 convs, 2x2/2 deconvolution, per-class "
                        "28x28 masks, <= 64 positive RoIs) + SCDA losses, synthetic 800x1344 + elliptical instance masks, batch=1/GPU "
                        "(BASELINE.json configs[4]; the reference's model file is missing: performance only)"),
                       "image": [bh, bw], "gt_boxes": G, "rois": 512, **({"mask_rois": mask_rois} if mask_rois is not None else {}), "clusters": 4, "recon": tr.recon, "parallelism": "dp%d" % world,
                       "iters_per_s": round(world * a.steps / dt, 3), "final_loss": round(float(out["loss"]), 4),
                       "images_per_step": "1 source (forward + backward) + 1 target (forward only: it carries no loss), per GPU",
                       "proposals_post_nms": {"source": tr.last_num_proposals[0], "target": tr.last_num_proposals[1], "quota": quota},
                       "preconditioning_iterations": a.warmup + precond},
            "roofline": roof,
            # where the iteration's time is (verdict r05 item 2): device time by segment, and what the host spent
            "segments": segments,
            "host_ms_per_step": round(cpu_main / a.steps * 1e3, 3),
            "host": {"main_thread_cpu_ms_per_step": round(cpu_main / a.steps * 1e3, 3),
                     "process_cpu_ms_per_step": round(cpu_proc / a.steps * 1e3, 3),
                     "note": "CPU time over the timed region / steps; the main thread enqueues the whole iteration and spins while it waits "
                             "for a device result (RPN outputs, keep lists), so main-thread time ~ wall time unless blocking waits are on",
                     "cpu_quota": _hostenv.cpu_quota(), "cpu_count": os.cpu_count()},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline() if a.config == "vgg16" else cpu_baseline_resnet(a.config == "maskrcnn")
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
