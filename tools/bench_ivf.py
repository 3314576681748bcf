#!/usr/bin/env python3
"""IVF-flat (the reference's optional index type) at the benchmark scale: 1M x 512, nlist 4096, nprobe 32 (config.py:76),
k = 15.  Prints build times, search time per batch, recall@15 against the exact flat search."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N, D, K, NLIST, NPROBE = 1_000_000, 512, 15, 4096, 32
# clustered store: 2000 gaussian blobs (uniform random data would make any IVF useless)
cent = torch.empty((2000, D), device=dev)
_lib.check(lib.radad_synth_rows(cent.data_ptr(), 0, 2000, D, 11, 0, _lib.stream_ptr(dev)))
rows = torch.empty((N, D), device=dev)
_lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, N, D, 4321, 0, _lib.stream_ptr(dev)))
rows += 3.0 * cent[(torch.arange(N, device=dev) * 7919) % 2000]
out = {}
idx = R.HipIVFFlatIndex(D, NLIST, 0)
t0 = time.perf_counter(); idx.train(rows[:50000]); torch.cuda.synchronize(); out["train_s"] = round(time.perf_counter() - t0, 3)
t0 = time.perf_counter(); idx.add(rows); torch.cuda.synchronize(); out["add_s"] = round(time.perf_counter() - t0, 3)
flat = R.HipFlatIndex(D, _lib.METRIC_L2, 0); flat.add_device(rows)
idx.nprobe = NPROBE
NQS = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 256, 1024)
for nq in NQS:
    q = rows[(torch.arange(nq, device=dev) * 977 + 5) % N] + 0.5 * torch.randn((nq, D), device=dev)
    for _ in range(2):
        Di, Ii = idx.search_device(q, K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        Di, Ii = idx.search_device(q, K)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
    De, Ie = flat.search_device(q, K)
    rec = float(np.mean([len(set(a) & set(b)) / K for a, b in zip(Ii.cpu().tolist(), Ie.cpu().tolist())]))
    out[f"nq{nq}"] = {"search_ms": round(ms, 3), "recall_at_15_vs_flat": round(rec, 4)}
print(json.dumps(out))
