"""iteration time with the whole RCCL choreography of a data-parallel run on ONE rank (async all-reduce of the four flat buckets on
RCCL's stream, waits, stream hand-overs; the collective itself moves nothing) against the plain step -- what the extra stream costs.
   [GPU_MAX_HW_QUEUES=n] python scripts/onerank_rccl_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist, bench
from scda_amd.train_step import ScdaTrainer
if os.environ.get("SCDA_BLOCKING_SYNC"):
    from scda_amd.hostenv import prefer_blocking_sync; prefer_blocking_sync(0)
dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1)
for coll in ([bool(int(sys.argv[1]))] if len(sys.argv) > 1 else (False, True)):   # ONE trainer per process: streams of deleted trainers stay
    torch.manual_seed(0); np.random.seed(100)
    tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H, collectives=coll)
    src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
    for _ in range(8): tr.step(src, gts, info, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.step(src, gts, info, tgt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("collectives=%-5s AB_STREAMS=%s SIDE=%s GRAPH=%s SEGMENTED=%s GPU_MAX_HW_QUEUES=%s %s: %.2f ms / iteration (host done enqueueing after %.2f)" % (coll, os.environ.get("SCDA_AB_STREAMS", "1"),
          os.environ.get("SCDA_SIDE_STREAM", "1"), "on" if tr._gan_graph_ok() else "off", os.environ.get("SCDA_SEGMENTED_REDUCE", "1"), os.environ.get("GPU_MAX_HW_QUEUES", "default") + (" BLOCKING" if os.environ.get("SCDA_BLOCKING_SYNC") else ""),
          " ".join("%s=%s" % kv for kv in os.environ.items() if kv[0].startswith(("TORCH_NCCL", "NCCL_", "RCCL_"))),
          (time.perf_counter() - t0) / 30 * 1e3, (t1 - t0) / 30 * 1e3), flush=True)
    del tr
dist.destroy_process_group()
