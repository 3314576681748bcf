"""GPU, 2 processes: the real chain of a sharded search -- HipFlatIndex(id_base) on every rank's shard, search_begin ->
all-reduce(max) of the per-query k-th-best bounds -> search_finish (or the one-piece search_device), exchange of the per-shard
float64 lists, radad_topk_merge_f64 -- must equal the oracle over the whole store.  On a box with >= 2 GPUs the ranks take one
GPU each and the collectives are RCCL ("nccl"); on a one-GPU box both ranks share cuda:0 and the collectives are gloo, staged
through the host (everything above the transport is exercised either way)."""
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


def _worker(rank, world, port, metric_name, n, nq_locals, dim, k, exchange, bounded, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    rccl = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if rccl else 0)
    torch.cuda.set_device(dev)
    if rccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import radad_oracle as O, synth
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import ShardedSearch, shard_bounds
    metric = {"L2": _lib.METRIC_L2, "COSINE": _lib.METRIC_COSINE}[metric_name]
    db = synth.rows(0, n, dim, 4321)
    starts = np.concatenate([[0], np.cumsum(nq_locals)])
    q_all = synth.rows(0, int(starts[-1]), dim, 977)
    for j in range(len(q_all)):                           # one planted neighbour per query, spread over the shards
        db[(j * 769 + 5) % n] = q_all[j] + np.float32(0.05) * synth.rows(j, 1, dim, 55)[0]
    lo, hi = shard_bounds(n, world, rank)
    idx = HipFlatIndex(dim, metric, dev.index, id_base=lo)
    idx.add(db[lo:hi])

    def local_search(q, kk):
        _, ids, key64 = idx.search_device(q, kk, return_f64=True)
        return key64, ids

    def finish(lb):
        _, ids, key64 = idx.search_finish(lb, return_f64=True)
        return key64, ids

    s = ShardedSearch(local_search, metric, uneven=len(set(nq_locals)) > 1, exchange=exchange,
                      bounded=(idx.search_begin, finish) if bounded else None)
    sl = slice(int(starts[rank]), int(starts[rank + 1]))
    d, i = s.search(torch.from_numpy(q_all[sl]).to(dev), k)
    da, ia = s.search(torch.from_numpy(q_all[sl]).to(dev), k, return_all=True)
    od, oi = O.knn(db, q_all, k, metric_name)
    ok = (np.array_equal(i.cpu().numpy(), oi[sl]) and np.array_equal(ia.cpu().numpy(), oi)
          and np.allclose(d.cpu().numpy(), od[sl], rtol=1e-5, atol=1e-5) and np.allclose(da.cpu().numpy(), od, rtol=1e-5, atol=1e-5)
          and tuple(d.shape) == (nq_locals[rank], k))
    if bounded and ok:
        # with the global bound a shard re-ranks only what can be among the global k best: fewer candidates than its own top k needs
        cert = idx.last_launch()["certificate"]
        ok = cert["rejected"] <= max(1, cert["queries"] // 50)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric,n,nq_locals,k,exchange,bounded", [
    ("COSINE", 70001, [150, 150], 10, "all_to_all", False),     # every shard on the certified f16 scan (>= 16384 rows, 300 queries)
    ("COSINE", 70001, [150, 150], 10, "all_to_all", True),      # ... in two halves around the bound exchange
    ("L2", 70001, [150, 150], 15, "all_gather", True),          # L2: the bound is on -|q - y|^2
    ("L2", 9001, [40, 3], 15, "all_gather", False),             # query counts differ per rank; fp32 tile kernel
    ("L2", 9001, [40, 3], 15, "all_to_all", True),              # ... which offers no bound (-inf): finish() returns the shard's own top k
])
def test_sharded_search_two_ranks_one_gpu(gpu, metric, n, nq_locals, k, exchange, bounded):
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), metric, n, nq_locals, 64, k, exchange, bounded, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
