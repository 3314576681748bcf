"""GPU, 2 processes on ONE MI355X, gloo collectives staged through the host: the real chain of a sharded search --
HipFlatIndex(id_base).search_device(return_f64) on every rank's shard -> exchange of the per-shard float64 lists ->
radad_topk_merge_f64 -- must equal the oracle over the whole store.  (The RCCL transport itself needs >= 2 GPUs: the
driver's multi-GPU bench is its first run; everything above the transport is exercised here.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, metric_name, n, nq_locals, dim, k, exchange, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import radad_oracle as O, synth
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import ShardedSearch, shard_bounds
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    metric = {"L2": _lib.METRIC_L2, "COSINE": _lib.METRIC_COSINE}[metric_name]
    db = synth.rows(0, n, dim, 4321)
    starts = np.concatenate([[0], np.cumsum(nq_locals)])
    q_all = synth.rows(0, int(starts[-1]), dim, 977)
    for j in range(len(q_all)):                           # one planted neighbour per query, spread over the shards
        db[(j * 769 + 5) % n] = q_all[j] + np.float32(0.05) * synth.rows(j, 1, dim, 55)[0]
    lo, hi = shard_bounds(n, world, rank)
    idx = HipFlatIndex(dim, metric, 0, id_base=lo)
    idx.add(db[lo:hi])

    def local_search(q, kk):
        _, ids, key64 = idx.search_device(q, kk, return_f64=True)
        return key64, ids

    s = ShardedSearch(local_search, metric, uneven=len(set(nq_locals)) > 1, exchange=exchange)
    sl = slice(int(starts[rank]), int(starts[rank + 1]))
    d, i = s.search(torch.from_numpy(q_all[sl]).to(dev), k)
    da, ia = s.search(torch.from_numpy(q_all[sl]).to(dev), k, return_all=True)
    od, oi = O.knn(db, q_all, k, metric_name)
    ok = (np.array_equal(i.cpu().numpy(), oi[sl]) and np.array_equal(ia.cpu().numpy(), oi)
          and np.allclose(d.cpu().numpy(), od[sl], rtol=1e-5, atol=1e-5) and np.allclose(da.cpu().numpy(), od, rtol=1e-5, atol=1e-5)
          and tuple(d.shape) == (nq_locals[rank], k))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric,n,nq_locals,k,exchange", [
    ("COSINE", 70001, [150, 150], 10, "all_to_all"),      # every shard on the certified f16 scan (>= 16384 rows, 300 queries)
    ("L2", 9001, [40, 3], 15, "all_gather"),              # query counts differ per rank; fp32 tile kernel
])
def test_sharded_search_two_ranks_one_gpu(gpu, metric, n, nq_locals, k, exchange):
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), metric, n, nq_locals, 64, k, exchange, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
