import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
lib = _lib.load()
gpu = torch.device("cuda:0")
n, dim, nq, k = 1_300_000, 384, 700, 10
rows = torch.empty((n, dim), device=gpu)
_lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, n, dim, 4321, 0, _lib.stream_ptr(gpu)))
q = torch.empty((nq, dim), device=gpu)
_lib.check(lib.radad_synth_rows(q.data_ptr(), 0, nq, dim, 977, 0, _lib.stream_ptr(gpu)))
planted = (torch.arange(nq, device=gpu) * 1801 + 29) % n
rows[planted] = q + 0.05 * rows[:nq]
for metric in (_lib.METRIC_COSINE, _lib.METRIC_L2):
    out = {}
    for live in (1, 0):
        idx = HipFlatIndex(dim, metric, 0, live_floor=live)
        idx.add_device(rows)
        for nqq in (700, 512, 256, 100):
            for rep in range(2):
                D, I = idx.search_device(q[:nqq].contiguous(), k)
                print(metric, "live", live, "nq", nqq, idx.last_launch(), flush=True)
            out[(live, nqq)] = I
        del idx
    for nqq in (700, 512, 256, 100):
        print("equal", nqq, torch.equal(out[(1, nqq)], out[(0, nqq)]))
