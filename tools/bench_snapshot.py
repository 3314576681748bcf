#!/usr/bin/env python3
"""Snapshot save / load rates: 1 M x 512 fp32 store (2.05 GB) to a file under $TMPDIR and back, whole and as
one of 8 row shards.  Page cache is warm for the loads (the file was just written)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib

N, D = 1_000_000, 512
dev = torch.device("cuda:0")
lib = _lib.load()
rows = torch.empty((N, D), device=dev)
_lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, N, D, 4321, 0, _lib.stream_ptr(dev)))
idx = R.HipFlatIndex(D, _lib.METRIC_COSINE, device=0)
idx.add_device(rows)
del rows
torch.cuda.synchronize()
out = {"rows": N, "dim": D, "bytes": N * D * 4}
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "store.radad")
    t = time.perf_counter(); idx.save(path); out["save_s"] = round(time.perf_counter() - t, 3)
    b = R.HipFlatIndex(D, _lib.METRIC_COSINE, device=0)
    for name in ("load_s", "load_again_s"):
        t = time.perf_counter(); b.load(path); out[name] = round(time.perf_counter() - t, 3)
    t = time.perf_counter(); s = R.HipFlatIndex.load_shard(path, 3, 8, device=0); out["load_shard_1of8_s"] = round(time.perf_counter() - t, 3)
    out["shard_rows"] = s.ntotal
    ids = torch.arange(0, N, 9973, device=dev)
    out["roundtrip_equal"] = bool(torch.equal(idx.reconstruct_batch(ids), b.reconstruct_batch(ids)))
out["save_GBps"] = round(out["bytes"] / out["save_s"] / 1e9, 2)
out["load_GBps"] = round(out["bytes"] / out["load_again_s"] / 1e9, 2)
print(json.dumps(out))
