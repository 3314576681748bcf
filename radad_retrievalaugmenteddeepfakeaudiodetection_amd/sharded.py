"""Row-sharded retrieval across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-GPU (vector_database.py:23 `device_id = 0`).  The store shards naturally: rank r owns a
contiguous row range [base_r, base_r + n_r) and reports GLOBAL ids (id_base = base_r).  A search is
  1. all_gather the per-rank query blocks            [Q_r, D] -> [Q, D]            (embeds are data-parallel)
  2. local brute-force top-k of ALL queries on the shard                            (no communication)
  3. all_gather the per-shard (dist, id) lists       2 x [Q, k] per rank          (12*Q*k bytes per rank: latency bound)
  4. merge the G sorted lists per query by (distance, id)                          (radad_topk_merge)
Both collectives are tiny, so on a fully connected xGMI node they are one-hop all-gathers.
"""
import ctypes as C
from typing import Callable, Optional, Tuple

import numpy as np

from . import _lib


def shard_bounds(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range of `rank`: the first n_total % world_size ranks get one extra row."""
    q, r = divmod(int(n_total), int(world_size))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def hip_merge(metric: int, dists, idxs, k: int):
    """dists/idxs: [G, Q, k] CUDA tensors -> merged ([Q,k] f32, [Q,k] i64).  float64 `dists` (the keys
    radad_knn_search_f64 returns) are merged on float64 so no cross-shard pair is decided by fp32 rounding."""
    import torch
    lib = _lib.load()
    G, Q, kk = dists.shape
    assert kk == k
    out_d = torch.empty((Q, k), device=dists.device, dtype=torch.float32)
    out_i = torch.empty((Q, k), device=dists.device, dtype=torch.int64)
    d, i = dists.contiguous(), idxs.contiguous()
    with torch.cuda.device(dists.device):
        if d.dtype == torch.float64:
            _lib.check(lib.radad_topk_merge_f64(metric, d.data_ptr(), i.data_ptr(), G, Q, k, out_d.data_ptr(), out_i.data_ptr(),
                                                None, dists.device.index, _lib.stream_ptr(dists.device)), "radad_topk_merge_f64")
        else:
            _lib.check(lib.radad_topk_merge(metric, d.float().data_ptr() if d.dtype != torch.float32 else d.data_ptr(),
                                            i.data_ptr(), G, Q, k, out_d.data_ptr(), out_i.data_ptr(), dists.device.index,
                                            _lib.stream_ptr(dists.device)), "radad_topk_merge")
    return out_d, out_i


class ShardedSearch:
    """Collective search over per-rank shards.

    local_search(q [Q,D], k) -> (dist [Q,k], gid [Q,k]) must return GLOBAL ids (HipFlatIndex with id_base does);
    dist may be float64 (HipFlatIndex.search_device(..., return_f64=True)[2]) -- it is what gets merged.
    merge(metric, dists [G,Q,k], idxs [G,Q,k], k) -> ([Q,k],[Q,k]); defaults to the HIP merge kernel.
    Every rank must call `search` with the same k (and the same number of local queries unless
    `uneven=True`, which pads to the max).
    """

    def __init__(self, local_search: Callable, metric: int, group=None, merge: Optional[Callable] = None):
        import torch.distributed as dist
        self.local_search = local_search
        self.metric = int(metric)
        self.group = group
        self.merge = merge or hip_merge
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def _all_gather(self, t):
        """[n, ...] per rank -> [world*n, ...] (rank-major).  RCCL gathers device tensors in place; a gloo group
        (CPU tests, or rehearsing several ranks on one GPU) stages through host memory."""
        import torch
        import torch.distributed as dist
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            host = torch.empty(out.shape, dtype=t.dtype)
            dist.all_gather_into_tensor(host, t.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def _exchange(self, t, qr: int):
        """t [world*qr, k]: this shard's lists for ALL queries, rank-major -> [world, qr, k]: every shard's lists for THIS
        rank's queries.  One all-to-all (each rank receives only the rows it merges: 1/world of an all-gather's traffic);
        backends without it fall back to the all-gather and a slice."""
        import torch
        import torch.distributed as dist
        t = t.contiguous()
        k = t.shape[1]
        try:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                host_out = torch.empty((self.world * qr, k), dtype=t.dtype)
                dist.all_to_all_single(host_out, t.cpu(), group=self.group)
                return host_out.to(t.device).view(self.world, qr, k)
            out = torch.empty((self.world * qr, k), device=t.device, dtype=t.dtype)
            dist.all_to_all_single(out, t, group=self.group)
            return out.view(self.world, qr, k)
        except (RuntimeError, NotImplementedError):
            Q = t.shape[0]
            sl = slice(self.rank * qr, (self.rank + 1) * qr)
            return self._all_gather(t).view(self.world, Q, k)[:, sl].contiguous()

    def gather_queries(self, q_local):
        return q_local if self.world == 1 else self._all_gather(q_local)

    def search(self, q_local, k: int, return_all: bool = False):
        """q_local [Q_r, D] (same Q_r on every rank) -> this rank's rows of the merged result
        ([Q_r,k] distances, [Q_r,k] global ids); `return_all` returns all Q rows instead."""
        import torch
        import torch.distributed as dist
        q_all = self.gather_queries(q_local)
        d_loc, i_loc = self.local_search(q_all, k)
        if self.world == 1:
            return d_loc.float(), i_loc
        Q = q_all.shape[0]
        if return_all:
            # concatenated output form (accepted by both RCCL and gloo), viewed as [G, Q, k]
            d_all = self._all_gather(d_loc).view(self.world, Q, k)
            i_all = self._all_gather(i_loc).view(self.world, Q, k)
            return self.merge(self.metric, d_all, i_all, k)
        qr = q_local.shape[0]
        return self.merge(self.metric, self._exchange(d_loc, qr), self._exchange(i_loc, qr), k)
