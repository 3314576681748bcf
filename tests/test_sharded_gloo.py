"""CPU, 2-8 processes, gloo: the collective structure of the row-sharded search (partition, global ids, all-gather of
queries, all-to-all / all-gather of per-shard lists, merge, each rank keeping its own query rows; uneven row counts, query
counts that differ per rank, k larger than a shard).  The local search and the merge are the ORACLE here (there is no GPU
in this test); on the GPU box tests/test_gpu_sharded.py runs the same ShardedSearch with HipFlatIndex.search_device and
radad_topk_merge_f64 in two processes on one GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, metric, n, nq_locals, dim, k, exchange, out, bounded=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import radad_oracle as O, synth
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import ShardedSearch, shard_bounds
    db = synth.rows(0, n, dim, 4321)
    lo, hi = shard_bounds(n, world, rank)
    shard = db[lo:hi]

    def local_search(q, kk):                      # oracle stand-in for HipFlatIndex(id_base=lo).search_device(return_f64)
        d, i = O.knn(shard, q.numpy(), kk, metric) if len(shard) else (np.zeros((len(q), 0)), np.zeros((len(q), 0), np.int64))
        pad = kk - d.shape[1]
        if pad > 0:
            fill = np.inf if metric == "L2" else -np.inf
            d = np.concatenate([d, np.full((len(d), pad), fill)], 1)
            i = np.concatenate([i, np.full((len(i), pad), -1 - lo, np.int64)], 1)
        return torch.from_numpy(d), torch.from_numpy(i + lo)

    def merge(m, d, i, kk):
        md, mi = O.merge_topk(list(d.numpy()), list(i.numpy()), kk, metric)
        return torch.from_numpy(md), torch.from_numpy(mi)

    # oracle stand-ins for HipFlatIndex.search_begin / search_finish: a lower bound of the shard's exact k-th best SCORE (larger is
    # better: inner product, or minus the squared distance), then only the rows that reach the best bound any shard has
    state = {}

    def begin(q, kk):
        d, i = local_search(q, kk)
        sc = -d if metric == "L2" else d
        state["res"] = (d, i, sc)
        if bounded == "topk":      # the library's form: bounds of the k best rows, -inf where the shard has fewer
            return torch.nan_to_num(sc.float(), nan=-float("inf"), neginf=-float("inf")).clone()
        return sc[:, kk - 1].float().clone() if kk <= len(shard) else torch.full((len(q),), -float("inf"))

    def finish(glb):
        d, i, sc = state.pop("res")
        # (the bound travels as float32: compare against it rounded DOWN to stay a lower bound)
        g = torch.nextafter(glb.double().float(), torch.tensor(-float("inf"))).double()[:, None]
        drop = sc < g
        d = torch.where(drop, torch.full_like(d, float("inf") if metric == "L2" else -float("inf")), d)
        i = torch.where(drop, torch.full_like(i, -1), i)
        return d, i

    uneven = len(set(nq_locals)) > 1
    s = ShardedSearch(local_search, 0 if metric == "L2" else 1, merge=merge, uneven=uneven, exchange=exchange,
                      bounded=(begin, finish) if bounded else None)
    assert (s.world, s.rank) == (world, rank)
    starts = np.concatenate([[0], np.cumsum(nq_locals)])
    q_all = synth.rows(0, int(starts[-1]), dim, 977)
    sl = slice(int(starts[rank]), int(starts[rank + 1]))
    q_local = torch.from_numpy(q_all[sl])
    d, i = s.search(q_local, k)
    da, ia = s.search(q_local, k, return_all=True)
    od, oi = O.knn(db, q_all, k, metric)
    if od.shape[1] < k:                                  # fewer rows than k: unfilled slots are id -1 / +-inf, as faiss
        pad = k - od.shape[1]
        od = np.concatenate([od, np.full((len(od), pad), np.inf if metric == "L2" else -np.inf)], 1)
        oi = np.concatenate([oi, np.full((len(oi), pad), -1, np.int64)], 1)
    # (float64 sums over a shard and over the whole store may differ in the last bit: BLAS blocks them differently)
    close = lambda a, b: a.shape == b.shape and np.allclose(a, b, rtol=1e-12, atol=1e-12)
    ok = (np.array_equal(i.numpy(), oi[sl]) and close(d.numpy(), od[sl]) and np.array_equal(ia.numpy(), oi)
          and close(da.numpy(), od) and d.shape == (nq_locals[rank], k))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,metric,n,nq_locals,k,exchange", [
    (2, "L2", 1001, [5, 5], 7, "all_to_all"),
    (2, "IP", 64, [3, 3], 10, "all_gather"),
    (2, "L2", 3, [2, 2], 4, "all_to_all"),              # k larger than the whole store: -1 / inf padding survives the merge
    (4, "L2", 1003, [4, 4, 4, 4], 6, "all_to_all"),     # n % world != 0
    (4, "IP", 45, [3, 0, 5, 1], 15, "all_to_all"),      # query counts differ per rank (one rank has none); k > shard rows (11)
    (8, "L2", 2005, [2, 3, 1, 2, 2, 4, 2, 2], 5, "all_to_all"),
    (8, "IP", 6, [1] * 8, 3, "all_gather"),             # fewer rows than ranks: two shards are empty
])
def test_sharded_search(world, metric, n, nq_locals, k, exchange):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), metric, n, nq_locals, 16, k, exchange, out), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}


@pytest.mark.parametrize("world,metric,n,nq_locals,k,exchange,form", [
    (2, "L2", 1001, [5, 5], 7, "all_to_all", "topk"),
    (4, "IP", 45, [3, 0, 5, 1], 15, "all_to_all", "topk"),     # k > shard rows: a shard's missing bounds are -inf, the union still has k
    (4, "IP", 45, [3, 0, 5, 1], 15, "all_to_all", "kth"),      # the weaker one-value form (all-reduce max): such shards offer nothing
    (8, "L2", 2005, [2, 3, 1, 2, 2, 4, 2, 2], 5, "all_gather", "topk"),
])
def test_sharded_search_with_a_global_bound(world, metric, n, nq_locals, k, exchange, form):
    """the two-phase form (search_begin -> all-reduce(max) of the k-th-best bounds -> search_finish): shards return only rows that
    can be among the global k best (-1 filled otherwise); the merged result must still be the unsharded one"""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), metric, n, nq_locals, 16, k, exchange, out, form), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}
