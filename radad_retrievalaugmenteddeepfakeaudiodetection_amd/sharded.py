"""Row-sharded retrieval across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference is single-GPU (vector_database.py:23 `device_id = 0`).  The store shards naturally: rank r owns a
contiguous row range [base_r, base_r + n_r) and reports GLOBAL ids (id_base = base_r).  A search is
  1. all_gather the per-rank query blocks            [Q_r, D] -> [Q, D]            (embeds are data-parallel)
  2. local brute-force top-k of ALL queries on the shard                            (no communication)
  3. all_gather the per-shard (dist, id) lists       2 x [Q, k] per rank          (12*Q*k bytes per rank: latency bound)
  4. merge the G sorted lists per query by (distance, id)                          (radad_topk_merge)
Both collectives are tiny, so on a fully connected xGMI node they are one-hop all-gathers.

With `bounded=(begin, finish)` (HipFlatIndex.search_begin / search_finish) step 2 is split around one more tiny collective:
  2a. scan the shard; per query lower bounds of the exact scores of its k best rows          (radad_knn_search_begin)
  2b. all_gather of the bounds                       [Q, k] float32 per rank (4*Q*k bytes); the k-th largest of a query's G*k
      values is a lower bound of the exact k-th best score of the WHOLE store
  2c. float64 re-rank of only those candidates that can still be among the GLOBAL k best      (radad_knn_search_finish)
Without it every shard certifies ITS OWN top k: on G shards the node re-ranks G times what one GPU would (rehearsed at G = 8:
143 candidates per query and shard against 151 per query on one GPU), and the re-rank does not scale.
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
    Every rank must call `search` with the same k.  uneven=False (default): every rank passes the same number of local
    queries.  uneven=True: the counts may differ (or be zero); one extra tiny all-gather of the counts per search, local
    blocks are padded to the largest.
    exchange: how the per-shard lists travel -- "all_to_all" (each rank receives only the lists of its own queries: 1/world
    of an all-gather's traffic; RCCL and gloo both implement it) or "all_gather" (+ slice).  It is fixed HERE, identically on
    every rank: a collective is never retried with a different primitive (a rank that failed alone would leave the others
    inside the first one).
    """

    def __init__(self, local_search: Callable, metric: int, group=None, merge: Optional[Callable] = None,
                 uneven: bool = False, exchange: str = "all_to_all", bounded=None, timing: bool = False):
        import torch.distributed as dist
        if exchange not in ("all_to_all", "all_gather"):
            raise ValueError("exchange must be 'all_to_all' or 'all_gather'")
        self.local_search = local_search
        self.metric = int(metric)
        self.group = group
        self.merge = merge or hip_merge
        self.uneven = bool(uneven)
        self.exchange = exchange
        # bounded = (begin, finish[, abort]): begin(q [Q,D], k) -> float32 [Q, k] lower bounds of the exact scores of this shard's k best
        # rows (or [Q]: of its k-th best alone); finish(global_lb [Q] or None) -> (dist [Q,k] (float64 keys), gid [Q,k]), rows short of
        # k are -1 filled; abort() gives the begun search up when the exchange fails (without it finish(None) is called).  Same on
        # every rank.
        self.bounded = bounded
        self.timing = bool(timing)       # record (collective_ms, rerank_ms) of every search (CUDA events; read with timings())
        self._events = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # gloo moves host memory: device tensors are staged through the host (CPU tests; rehearsing ranks on one GPU)
        self.staged = dist.is_initialized() and dist.get_backend(group) == "gloo"

    def _all_gather(self, t):
        """[n, ...] per rank -> [world*n, ...] (rank-major)."""
        import torch
        import torch.distributed as dist
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        if t.is_cuda and self.staged:
            host = torch.empty(out.shape, dtype=t.dtype)
            dist.all_gather_into_tensor(host, t.cpu(), group=self.group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def _exchange(self, t, qr: int):
        """t [world*qr, k]: this shard's lists for ALL queries, rank-major -> [world, qr, k]: every shard's lists for THIS
        rank's queries."""
        import torch
        import torch.distributed as dist
        t = t.contiguous()
        k = t.shape[1]
        if self.exchange == "all_gather":
            sl = slice(self.rank * qr, (self.rank + 1) * qr)
            return self._all_gather(t).view(self.world, self.world * qr, k)[:, sl].contiguous()
        if t.is_cuda and self.staged:
            host_out = torch.empty((self.world * qr, k), dtype=t.dtype)
            dist.all_to_all_single(host_out, t.cpu(), group=self.group)
            return host_out.to(t.device).view(self.world, qr, k)
        out = torch.empty((self.world * qr, k), device=t.device, dtype=t.dtype)
        dist.all_to_all_single(out, t, group=self.group)
        return out.view(self.world, qr, k)

    def _all_reduce_max(self, t):
        import torch.distributed as dist
        if t.is_cuda and self.staged:
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def timings(self):
        """[(collective_ms, rerank_ms)] of the searches since the last call (synchronises); needs timing=True and CUDA tensors"""
        out = []
        for ev in self._events:
            ev[-1].synchronize()
            coll = sum(a.elapsed_time(b) for a, b in ev[0])
            out.append((coll, ev[1][0].elapsed_time(ev[1][1]) if ev[1] else 0.0))
        self._events = []
        return out

    def gather_queries(self, q_local):
        return q_local if self.world == 1 else self._all_gather(q_local)

    def search(self, q_local, k: int, return_all: bool = False):
        """q_local [Q_r, D] -> this rank's rows of the merged result ([Q_r,k] distances, [Q_r,k] global ids);
        `return_all` returns the rows of all ranks' queries instead (rank-major; with uneven=True: padded blocks removed)."""
        import torch
        if self.world == 1:
            d_loc, i_loc = self.local_search(q_local, k)
            return d_loc.float(), i_loc
        qr = q_local.shape[0]
        counts = None
        if self.uneven:
            c = torch.tensor([qr], dtype=torch.int64, device=q_local.device)
            counts = [int(x) for x in self._all_gather(c).cpu().tolist()]
            qmax = max(max(counts), 1)
            if qr < qmax:                                   # pad with copies of a zero query: results are dropped below
                pad = torch.zeros((qmax - qr,) + tuple(q_local.shape[1:]), device=q_local.device, dtype=q_local.dtype)
                q_local = torch.cat([q_local, pad])
            qr_pad = qmax
        else:
            qr_pad = qr
        import contextlib
        rec = self.timing and q_local.is_cuda
        coll_ev, rr_ev = [], None

        @contextlib.contextmanager
        def timed(kind):
            if not rec:
                yield
                return
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            yield
            b.record()
            if kind == "c":
                coll_ev.append((a, b))
            else:
                nonlocal rr_ev
                rr_ev = (a, b)

        with timed("c"):
            q_all = self._all_gather(q_local)
        if self.bounded is not None:
            begin, finish = self.bounded[0], self.bounded[1]
            abort = self.bounded[2] if len(self.bounded) > 2 else None
            lb = begin(q_all, k)
            # Between begin and finish the shard holds a begun search: whatever goes wrong in the exchange (a collective that throws,
            # a bound kernel that refuses its shape), the second half must still run -- or be given up -- or every later search on
            # this index fails with "has not been finished".
            try:
                with timed("c"):
                    if lb.dim() == 2:      # the k best of every shard: the k-th largest of the union bounds the global k-th best
                        allb = self._all_gather(lb).view(self.world, lb.shape[0], lb.shape[1])
                        from .vector_database import HipFlatIndex
                        lb = HipFlatIndex.global_bound(allb, k)       # (torch.topk on the CPU or beyond the kernel's 1280 values per query)
                    else:
                        lb = self._all_reduce_max(lb)
            except BaseException:
                if abort is not None:
                    abort()
                else:
                    finish(None)                                      # (this shard's own top k: a valid, if unbounded, second half)
                raise
            with timed("r"):
                d_loc, i_loc = finish(lb)
        else:
            d_loc, i_loc = self.local_search(q_all, k)
        Q = q_all.shape[0]
        if return_all:
            d_all = self._all_gather(d_loc).view(self.world, Q, k)
            i_all = self._all_gather(i_loc).view(self.world, Q, k)
            md, mi = self.merge(self.metric, d_all, i_all, k)
            if counts is not None:
                keep = torch.cat([torch.arange(r * qr_pad, r * qr_pad + c) for r, c in enumerate(counts)]).to(md.device)
                md, mi = md[keep], mi[keep]
            return md, mi
        with timed("c"):
            xd, xi = self._exchange(d_loc, qr_pad), self._exchange(i_loc, qr_pad)
        md, mi = self.merge(self.metric, xd, xi, k)
        if rec:
            self._events.append((coll_ev, rr_ev, coll_ev[-1][1]))
        return md[:qr], mi[:qr]


class ReplicatedSearch:
    """The other way to use G GPUs (SURVEY 8e's tuning option; the reference is single-GPU, vector_database.py:23): every rank holds
    the WHOLE store and searches only its own queries -- no collective in the data path, a rank's step is the one-GPU step.  Every
    BASELINE store fits one 288 GB MI355X (1 M x 512: 3 GB with its f16 plane; 10 M x 512: 30 GB; 50 M x 256 fp16: 26 GB); sharding
    (ShardedSearch, north_star's partitioning) is for stores that do not, and costs every rank the scan of ALL queries plus G re-ranks.
    Same call surface as ShardedSearch: search(q_local, k) -> this rank's rows; return_all=True gathers every rank's rows (rank-major;
    the only collective, and not part of a search)."""

    def __init__(self, local_search: Callable, group=None):
        import torch.distributed as dist
        self.local_search = local_search
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.timing = False
        self._gather = ShardedSearch(None, 0, group=group)._all_gather

    def timings(self):
        return []

    def search(self, q_local, k: int, return_all: bool = False):
        d, i = self.local_search(q_local, k)
        d = d.float()
        if not return_all or self.world == 1:
            return d, i
        return self._gather(d), self._gather(i)
