"""CPU, 2 processes, gloo: the collective structure of the row-sharded search (partition, global ids, all-gather of
queries and of per-shard lists, merge, each rank keeping its own query rows).  The local search and the merge are the
ORACLE here (there is no GPU in this test); on the GPU box the same ShardedSearch runs with HipFlatIndex.search_device
and radad_topk_merge_f64 (tests/test_gpu_knn.py::test_knn_id_base_and_merge_equals_unsharded covers those)."""
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


def _worker(rank, world, port, metric, n, nq_local, dim, k, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import radad_oracle as O, synth
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import ShardedSearch, shard_bounds
    db = synth.rows(0, n, dim, 4321)
    lo, hi = shard_bounds(n, world, rank)
    shard = db[lo:hi]

    def local_search(q, kk):                      # oracle stand-in for HipFlatIndex(id_base=lo).search_device(return_f64)
        d, i = O.knn(shard, q.numpy(), kk, metric)
        pad = kk - d.shape[1]
        if pad > 0:
            fill = np.inf if metric == "L2" else -np.inf
            d = np.concatenate([d, np.full((len(d), pad), fill)], 1)
            i = np.concatenate([i, np.full((len(i), pad), -1 - lo, np.int64)], 1)
        return torch.from_numpy(d), torch.from_numpy(i + lo)

    def merge(m, d, i, kk):
        md, mi = O.merge_topk(list(d.numpy()), list(i.numpy()), kk, metric)
        return torch.from_numpy(md), torch.from_numpy(mi)

    s = ShardedSearch(local_search, 0 if metric == "L2" else 1, merge=merge)
    assert (s.world, s.rank) == (world, rank)
    q_all = synth.rows(0, world * nq_local, dim, 977)
    q_local = torch.from_numpy(q_all[rank * nq_local:(rank + 1) * nq_local])
    d, i = s.search(q_local, k)
    da, ia = s.search(q_local, k, return_all=True)
    od, oi = O.knn(db, q_all, k, metric)
    if od.shape[1] < k:                                  # fewer rows than k: unfilled slots are id -1 / +-inf, as faiss
        pad = k - od.shape[1]
        od = np.concatenate([od, np.full((len(od), pad), np.inf if metric == "L2" else -np.inf)], 1)
        oi = np.concatenate([oi, np.full((len(oi), pad), -1, np.int64)], 1)
    sl = slice(rank * nq_local, (rank + 1) * nq_local)
    ok = (np.array_equal(i.numpy(), oi[sl]) and np.array_equal(d.numpy(), od[sl]) and np.array_equal(ia.numpy(), oi)
          and np.array_equal(da.numpy(), od) and d.shape == (nq_local, k))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric,n,nq_local,k", [("L2", 1001, 5, 7), ("IP", 64, 3, 10), ("L2", 3, 2, 4)])
def test_sharded_search_world2(metric, n, nq_local, k):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), metric, n, nq_local, 16, k, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
