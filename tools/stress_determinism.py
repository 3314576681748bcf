#!/usr/bin/env python3
"""Run the benchmark's step (embed 1024 synthetic clips, top-10 against a 1 M x 512 cosine store with planted neighbours) N times on
the same inputs and compare every step's embeddings, ids and distances BITWISE with the first step's: nothing in the chain may
depend on timing (LDS-DMA landing, hand-counted load queues, atomics in slot buffers, per-CU scheduling).
usage (through gpurun): python tools/stress_determinism.py --steps 400"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--metric", default="IP", choices=["IP", "L2"])
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
B, S, N, D, K = 1024, 64000, 1_000_000, 512, 10
cfg = R.Config()
cfg.update(device=dev, tpp_levels=[1], tpp_pooling_type="max", feature_dim=D, vector_db_index_type=a.metric)
fe = R.MelProjectionFeatureExtractor(cfg)
wave = torch.empty(B * S, device=dev)
_lib.check(lib.radad_synth_audio(wave.data_ptr(), 0, B, S, 1234, 0, _lib.stream_ptr(dev)))
offs = np.arange(B + 1, dtype=np.int64) * S
emb0 = fe.embed_clips(wave, offs).clone()
rows = torch.empty((N, D), device=dev)
_lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, N, D, 4321, 0, _lib.stream_ptr(dev)))
jj = torch.arange(B, device=dev)
rows[(jj * 977 + 17) % N] = emb0 + 0.05 * emb0.norm(dim=1, keepdim=True) / D ** 0.5 * torch.randn((B, D), device=dev, generator=torch.Generator(dev).manual_seed(7))
vdb = R.VectorDatabase(cfg)
vdb.create_index(D)
vdb.index.reserve(N)
vdb.index.add_device(rows)
del rows
D0, I0 = vdb.index.search_device(emb0, K)
D0, I0 = D0.clone(), I0.clone()
bad = {"embed": 0, "ids": 0, "dist": 0}
for s in range(a.steps):
    e = fe.embed_clips(wave, offs)
    Dd, Ii = vdb.index.search_device(e, K)
    bad["embed"] += int(not torch.equal(e, emb0))
    bad["ids"] += int(not torch.equal(Ii, I0))
    bad["dist"] += int(not torch.equal(Dd, D0))
    if s % 100 == 99:
        torch.cuda.synchronize()
        print(f"step {s + 1}: mismatching steps so far {bad}", flush=True)
torch.cuda.synchronize()
print({"steps": a.steps, "metric": a.metric, "mismatching_steps": bad, "certificate": vdb.index.last_launch()["certificate"]})
sys.exit(1 if any(bad.values()) else 0)
