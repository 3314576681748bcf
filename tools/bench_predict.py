#!/usr/bin/env python3
"""Small-batch (online `predict`, pipeline.py:1038-1054) scan timing: nq in {1, 4, 16} against 1M x 512, cosine, k = 15.
Prints ms per search (HIP events around the scan kernel) and the algorithmic HBM rate 4*N*D / t."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N, D, K = 1_000_000, 512, 15
rows = torch.empty((N, D), device=dev)
_lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, N, D, 4321, 0, _lib.stream_ptr(dev)))
idx = R.HipFlatIndex(D, _lib.METRIC_COSINE, 0)
idx.add_device(rows)
out = {}
for nq in (1, 4, 16, 17, 128):
    q = torch.empty((nq, D), device=dev)
    _lib.check(lib.radad_synth_rows(q.data_ptr(), 0, nq, D, 977, 0, _lib.stream_ptr(dev)))
    for _ in range(3):
        idx.search_device(q, K)
    idx.profile(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        D_, I_ = idx.search_device(q, K)
    t1.record(); torch.cuda.synchronize()
    ms = float(np.mean(idx.profile_read()))
    idx.profile(False)
    out[f"nq{nq}"] = {"scan_kernel_ms": round(ms, 4), "search_total_ms": round(t0.elapsed_time(t1) / 20, 4),
                      "store_GBps": round(4.0 * N * D / (ms * 1e-3) / 1e9, 1), "launch": idx.last_launch()}
print(json.dumps(out))
