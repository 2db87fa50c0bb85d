"""Cluster-region generator -- the contract of functions/mask.py:183-237: k-means over RoI centres, then per cluster
the first `threshold` member RoIs (or a with-replacement resample when the cluster is smaller) are stacked into
[N_cluster, threshold, F].  The k-means is sklearn's (third-party, as in the reference: KMeans(n_clusters,
random_state=0)); what changes here is the data path: the RoI features never leave the MI355X -- only the 512 RoI
boxes (already on the host) feed the clustering, and the gather runs on the device.

As in the reference the result is a NEW leaf tensor: no gradient flows back into the detector through it
(functions/mask.py:202-203,234 round-trip through numpy)."""
import numpy as np
import torch

from scda_amd.dropin import backend


def _np(x):
    return backend.host_array(x)


def proposals_to_centers(proposals):
    """[N,>=5] (b,x1,y1,x2,y2) -> [N,2] (cx, cy)"""
    return np.stack([(proposals[:, 3] + proposals[:, 1]) / 2.0, (proposals[:, 4] + proposals[:, 2]) / 2.0], axis=1)


_POOLS_LIMITED = []


def _limit_host_pools_once():
    """Pin the BLAS / OpenMP pools that numpy, scipy and sklearn bring along to ONE thread, once, for the life of the
    process (torch's own pool is left alone).  The host-side work of the SCDA step is 512-point k-means, 12 000-element
    sorts and a few small reductions: on a 256-thread host the default pools only hurt -- waking or re-spawning 255
    worker threads costs 30-60 ms per parallel region (measured with scripts/stall_sampler.py: stalls inside
    sklearn's Lloyd loop, np.einsum, np.var and threadpoolctl's own set_num_threads), and their spin-waiting preempts the
    thread that feeds the GPU.  Results do not depend on the thread count except for the last bits of the float32
    k-means centres (see tests/test_host_functions.py)."""
    if _POOLS_LIMITED:
        return
    import os
    if os.environ.get('SCDA_NO_POOL_LIMIT'):
        _POOLS_LIMITED.append(None)
        return
    try:
        from threadpoolctl import ThreadpoolController
        ctl = ThreadpoolController()
        ctl.lib_controllers = [c for c in ctl.lib_controllers if "torch" not in (c.filepath or "")]
        _POOLS_LIMITED.append(ctl.limit(limits=1))   # kept alive, never restored
    except Exception:  # threadpoolctl missing: correctness is unaffected
        _POOLS_LIMITED.append(None)


def cluster_indices(proposals_np, N_cluster=4, threshold=128):
    """-> (index int64 [N_cluster, threshold] into the RoI list, centres float64 [N_cluster, 2])"""
    from sklearn.cluster import KMeans
    _limit_host_pools_once()   # 512 two-dimensional points: one thread
    km = KMeans(n_clusters=N_cluster, random_state=0).fit(proposals_to_centers(proposals_np))
    rows = []
    for c in range(N_cluster):
        member = np.where(km.labels_[:] == c)[0]
        if member.shape[0] < threshold:
            member = member[np.random.choice(member.shape[0], threshold, replace=True)]
        else:
            member = member[0:threshold]
        rows.append(member)
    return np.stack(rows, axis=0).astype(np.int64), km.cluster_centers_


def compute_cluster_targets(proposals, features, N_cluster=4, threshold=128):
    """proposals [N,>=5], features [N,F] -> (cluster features [N_cluster, threshold, F] (leaf), centres [N_cluster,2])"""
    idx, centres = cluster_indices(_np(proposals), N_cluster, threshold)
    if features.is_cuda:
        from scda_amd import native
        flat = native.upload(idx.reshape(-1), features.device)
    else:
        flat = torch.from_numpy(idx.reshape(-1))
    with torch.no_grad():
        gathered = features.detach().index_select(0, flat).view(N_cluster, threshold, features.shape[1]).contiguous()
    return gathered.float(), centres
