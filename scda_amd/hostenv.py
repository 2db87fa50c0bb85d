"""Host-thread hygiene for the process that feeds the GPU.

The SCDA iteration's host side is a single Python thread issuing ~1500 small launches plus a few ms of numpy; it needs
one core.  Left alone, torch's intra-op pool (one thread per logical CPU, 256 on the MI355X hosts) and the BLAS/OpenMP
pools of numpy/scipy/sklearn wake up for tiny parallel regions and spin; inside a container with a CFS quota
(`cpu.max`, 16 CPUs on the benchmark boxes) that burns the quota within the first milliseconds of every 100 ms period
and the kernel then freezes *all* threads -- including the one launching kernels -- until the period ends.  Measured
(scripts/cpu_burn.py): 128 torch threads -> 16.0 cores busy, 35 of 36 periods throttled, 108 ms/iteration with the GPU
idle a third of the time; 1-4 threads -> 1.2 cores busy, no throttling, 54.6 ms/iteration.
"""
import os

_done = []


def cpu_quota():
    """CPUs this cgroup may use per period (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited/unknown"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else max(1, int(q) // int(p))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, q // p)
    except Exception:
        return None


def configure_host_threads(torch_threads=None):
    """idempotent; torch intra-op threads -> 1 unless SCDA_TORCH_THREADS overrides; numpy/scipy/sklearn pools -> 1.
    Nothing on the host path of the iteration has a parallel region worth a second thread, and idle OpenMP workers spin:
    measured (scripts/host_cpu_use.py) 4 threads = 3.6 cores busy per rank, 1 thread = 1.7 cores, same 32 ms/iteration --
    with 8 ranks inside one 16-CPU quota that is the difference between throttling and not."""
    if _done:
        return
    _done.append(True)
    if os.environ.get("SCDA_KEEP_HOST_THREADS"):
        return
    import torch
    n = torch_threads or int(os.environ.get("SCDA_TORCH_THREADS", "0")) or 1
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    from .dropin.functions.mask import _limit_host_pools_once
    _limit_host_pools_once()


def prefer_blocking_sync(device_index=0):
    """Make host-side waits on the device (stream / event synchronise) sleep instead of spin: hipSetDeviceFlags(
    hipDeviceScheduleBlockingSync) on THIS rank's device.  Call before the process touches the GPU.  Worth it when several
    ranks share the host cores (1.7 -> 1.5 cores busy per rank, iteration time unchanged); returns the HIP status (0 = ok)
    or None if the runtime library is not loadable."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return None
    rc = int(hip.hipSetDevice(ctypes.c_int(int(device_index))))   # the flags apply to the calling thread's current device
    if rc:
        return rc
    rc = int(hip.hipSetDeviceFlags(ctypes.c_uint(4)))
    if rc == 0:
        _blocking_sync[0] = True
    return rc


_blocking_sync = [False]


def data_parallel_env(world_size):
    """Process environment of a rank of a data-parallel run; call BEFORE the process touches the GPU (the HIP runtime reads it when it
    initialises).  GPU_MAX_HW_QUEUES=8: the runtime's default is four hardware queues per process; a rank drives five streams
    (compute, box logic, target branch, the nets' B halves, RCCL's), two of them then share a queue, and a stream that waits for an
    event deep in the backlog of another -- the all-reduce launched from inside the detector backward -- holds up its queue-mate.
    Measured with the whole RCCL choreography of a step on a one-rank group (scripts/onerank_matrix.sh,
    profiles/r05_onerank_rccl.txt; the plain step is 18.1 - 18.8 ms on the same boxes): 26.0 - 26.8 ms per iteration with four
    queues, 18.3 - 19.3 with eight, 18.6 - 19.6 with six.  With eight queues the EAGER iteration is also as fast as the hipGraph
    replays are with four (18.1 / 18.6 vs 18.15 / 18.75 ms) -- data-parallel runs lose nothing by doing without the graphs, which
    do NOT work with eight queues (25.7 ms).
    Only a default is set: what the caller exported wins.  -> dict of the variables this call set."""
    done = {}
    if world_size > 1 and "GPU_MAX_HW_QUEUES" not in os.environ:
        os.environ["GPU_MAX_HW_QUEUES"] = done["GPU_MAX_HW_QUEUES"] = "8"
    return done


def wants_blocking_sync(world_size):
    """several ranks on one host: sleep instead of spinning while waiting for the device only when the container's CPU quota is tight
    -- a spinning rank keeps 1.9 cores busy, a sleeping one 1.45 (scripts/host_budget_8ranks.py: 8 ranks = 15.1 / 11.7 cores of a
    16-CPU quota), and a throttled CFS period stalls every rank; with the collectives' stream in the process the blocking flag
    costs 0 - 2.5 ms per iteration from run to run (18.4 - 20.9 ms, profiles/r05_onerank_rccl.txt), which is why it is no longer
    taken whenever there is more than one rank.  SCDA_BLOCKING_SYNC=0 / 1 overrides."""
    v = os.environ.get("SCDA_BLOCKING_SYNC")
    if v is not None:
        return v not in ("", "0")
    if world_size <= 1:
        return False
    q = cpu_quota()
    return q is not None and q < 2.25 * world_size


def blocking_sync_selected():
    """prefer_blocking_sync() took effect in this process.  The trainer then keeps the GAN phases eager: under any non-default
    scheduling flag each hipGraph launch costs the device ~1.3 ms inside the runtime (profiles/r03_hipgraph.txt; round 4, same
    measurement: 22.4 ms per iteration with the graphs against 19.2 ms without, scripts/host_budget_8ranks.py)."""
    return _blocking_sync[0]
