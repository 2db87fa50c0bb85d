"""bench.py's N > 1 branch end to end (rendezvous, broadcast, barriers, MAX-over-ranks timing, one JSON line from rank 0),
launched exactly as the driver does -- `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` -- but with
both ranks on the single GPU of the test box and gloo carrying the collectives (SCDA_BENCH_DEVICE / SCDA_BENCH_BACKEND)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("how", ["torchrun", "plain"])
def test_bench_two_ranks(cuda, how):
    """`torchrun`: as the driver's documented N > 1 command.  `plain`: `python bench.py --gpus 2` by itself -- bench.py then starts
    its own ranks (one process per GPU, as the reference's launcher does)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SCDA_BENCH_DEVICE="0", SCDA_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2"]
    if how == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3      # whole-job images/s
    assert d["roofline"]["launches"] > 0
    assert d["collective"]["ranks"] == 2 and d["collective"]["backend"] == "gloo"
    # where the time is: device time between the trainer's phase marks (events in the marked iterations) and the host's CPU time per step
    seg = d["segments"]
    assert seg["step_begin->end"] > 0 and all(k in seg for k in ("backbones_enqueued", "det_backward_enqueued", "phase2", "phase3", "phase4+det_step"))
    assert abs(sum(v for k, v in seg.items() if k not in ("unit", "step_begin->end")) - seg["step_begin->end"]) <= 0.05 * seg["step_begin->end"]
    assert d["host_ms_per_step"] > 0 and d["host"]["process_cpu_ms_per_step"] >= d["host"]["main_thread_cpu_ms_per_step"] * 0.5
