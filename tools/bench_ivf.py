#!/usr/bin/env python3
"""IVF-flat (the reference's optional index type, vector_database.py:65-70,174-181) at the benchmark scale: 1M x 512, nlist 4096,
nprobe 32 (config.py:76), k = 15.  Prints build times and, per batch size, the search time, recall@15 against the exact flat
search, the flat search's own time, and a roofline block: the bytes of the lists the batch touches (f16 plane rows + their scale /
bias, each list once) over the search time against HBM (8 TB/s).  tools/bench_ivf.py [nq,nq,...] [hi_scan 0|1]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N, D, K, NLIST, NPROBE = 1_000_000, 512, 15, 4096, 32
NQS = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 256, 1024)
HI = int(sys.argv[2]) if len(sys.argv) > 2 else None
# clustered store: 2000 gaussian blobs (uniform random data would make any IVF useless)
cent = torch.empty((2000, D), device=dev)
_lib.check(lib.radad_synth_rows(cent.data_ptr(), 0, 2000, D, 11, 0, _lib.stream_ptr(dev)))
rows = torch.empty((N, D), device=dev)
_lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, N, D, 4321, 0, _lib.stream_ptr(dev)))
rows += 3.0 * cent[(torch.arange(N, device=dev) * 7919) % 2000]
out = {"config": {"rows": N, "dim": D, "k": K, "nlist": NLIST, "nprobe": NPROBE}}
idx = R.HipIVFFlatIndex(D, NLIST, 0, hi_scan=HI)
t0 = time.perf_counter(); idx.train(rows[:50000]); torch.cuda.synchronize(); out["train_s"] = round(time.perf_counter() - t0, 3)
t0 = time.perf_counter(); idx.add(rows); torch.cuda.synchronize(); out["add_s"] = round(time.perf_counter() - t0, 3)
flat = R.HipFlatIndex(D, _lib.METRIC_L2, 0); flat.add_device(rows)
idx.nprobe = NPROBE
assign = torch.from_numpy(idx.assignments()).to(dev)
list_rows = torch.bincount(assign.long(), minlength=NLIST)
centroids = torch.from_numpy(idx.centroids()).to(dev)


def timed(fn, reps=20):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps, r


for nq in NQS:
    q = rows[(torch.arange(nq, device=dev) * 977 + 5) % N] + 0.5 * torch.randn((nq, D), device=dev)
    ms, (Di, Ii) = timed(lambda: idx.search_device(q, K))
    info = idx.last_search_info()
    ms_flat, (De, Ie) = timed(lambda: flat.search_device(q, K))
    rec = float(np.mean([len(set(a) & set(b)) / K for a, b in zip(Ii.cpu().tolist(), Ie.cpu().tolist())]))
    # the lists this batch probes (float64 centroid distances, as the oracle's ivf_search): each is streamed once per task of <= 16 queries
    d2 = torch.cdist(q.double(), centroids.double())
    probes = d2.topk(NPROBE, largest=False).indices
    touched = torch.unique(probes)
    per_list_q = torch.bincount(probes.flatten(), minlength=NLIST)
    tasks = int(((per_list_q + 15) // 16).sum().item())
    row_bytes = (2 if info["scan"] == "hi_lists" else 4) * D + 8
    alg = int(list_rows[touched].sum().item()) * row_bytes
    streamed = int((list_rows * ((per_list_q + 15) // 16)).sum().item()) * row_bytes
    out[f"nq{nq}"] = {"search_ms": round(ms, 4), "flat_search_ms": round(ms_flat, 4), "recall_at_15_vs_flat": round(rec, 4), **info,
                      "lists_touched": int(touched.numel()), "tasks": tasks,
                      "roofline": {"bound": "hbm", "algorithmic_bytes": alg, "streamed_bytes": streamed, "achieved": round(alg / (ms * 1e-3) / 1e9, 1),
                                   "peak": 8000.0, "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / 1e9 / 8000.0, 4),
                                   "note": "whole search (coarse + grouping + list scan + re-rank) over the bytes of the touched lists"}}
print(json.dumps(out))
