#!/usr/bin/env python3
"""Scan-kernel timing for one (rows, dim, queries, store dtype) point: HIP-event durations of the scan kernel."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--nq", type=int, default=1024)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--f16", type=int, default=0)
ap.add_argument("--metric", default="cosine")
ap.add_argument("--reps", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
metric = {"cosine": _lib.METRIC_COSINE, "l2": _lib.METRIC_L2, "ip": _lib.METRIC_IP}[a.metric]
idx = R.HipFlatIndex(a.dim, metric, device=0, store_f16=bool(a.f16))
step = 1 << 18
for r0 in range(0, a.rows, step):
    n = min(step, a.rows - r0)
    rows = torch.empty((n, a.dim), device=dev)
    _lib.check(lib.radad_synth_rows(rows.data_ptr(), r0, n, a.dim, 4321, 0, _lib.stream_ptr(dev)))
    idx.add_device(rows)
q = torch.empty((a.nq, a.dim), device=dev)
_lib.check(lib.radad_synth_rows(q.data_ptr(), 0, a.nq, a.dim, 99, 0, _lib.stream_ptr(dev)))
idx.profile(True)
for _ in range(2):
    idx.search_device(q, a.k)
idx.profile_read()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.reps):
    D, I = idx.search_device(q, a.k)
t1.record(); torch.cuda.synchronize()
ms = idx.profile_read()
scan = sum(ms) / max(len(ms), 1) * idx.last_launch()["scan_launches"]
flops = 2.0 * a.nq * a.rows * a.dim
print(json.dumps({"rows": a.rows, "dim": a.dim, "nq": a.nq, "f16": a.f16, "metric": a.metric, "scan_ms": round(scan, 4),
                  "search_ms": round(t0.elapsed_time(t1) / a.reps, 4), "scan_TFLOPs": round(flops / scan / 1e9, 1),
                  "plane": idx.plane_info(), "launch": idx.last_launch()}))
