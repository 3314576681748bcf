"""Diagnosis (not product): candidates emitted per query and per query tile by the tile scan on a config-5-sized store, both scan forms
(radad_knn_last_emitted) -- how the round dependence of the first one-launch form was found.  usage: python tools/diag_emitted.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
lib = _lib.load()
gpu = torch.device("cuda:0")
n, dim, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000, 256, 10240, 10
q = torch.empty((nq, dim), device=gpu)
_lib.check(lib.radad_synth_rows(q.data_ptr(), 0, nq, dim, 977, 0, _lib.stream_ptr(gpu)))
planted = (torch.arange(nq, device=gpu) * (n // nq - 3) + 5) % n
for live in (1, 0):
    idx = HipFlatIndex(dim, _lib.METRIC_COSINE, 0, store_f16=True, live_floor=live)
    step = 1 << 20
    for r0 in range(0, n, step):
        m = min(step, n - r0)
        rows = torch.empty((m, dim), device=gpu)
        _lib.check(lib.radad_synth_rows(rows.data_ptr(), r0, m, dim, 4321, 0, _lib.stream_ptr(gpu)))
        sel = (planted >= r0) & (planted < r0 + m)
        ns = int(sel.sum())
        if ns:
            rows[planted[sel] - r0] = q[sel] + 0.05 * rows[:ns]
        idx.add_device(rows)
    del rows
    qb = q.to(torch.bfloat16)
    for rep in range(2):
        D, I = idx.search_device(qb, k)
        info = idx.last_launch()
        c, f = idx.last_emitted(nq)
        per_tile = c.reshape(-1, 256)
        print("live", live, info["scan_launches"], info["scan_phases"], info["certificate"], flush=True)
        print("  emitted: mean %.1f max %d; over 1024: %d; per query tile max:" % (c.mean(), c.max(), (c > 1024).sum()), per_tile.max(1).tolist(), flush=True)
        print("  per query tile mean:", np.round(per_tile.mean(1)).astype(int).tolist(), flush=True)
        print("  floors: min %.4f mean %.4f" % (f.min(), f.mean()), "over-1024 queries (first 20):", np.nonzero(c > 1024)[0][:20].tolist(), flush=True)
    del idx
