#!/usr/bin/env python3
"""What ONE rank of an N-GPU `bench.py --gpus N` run computes, reproduced in a single process on one GPU (the other ranks'
query blocks are embedded locally instead of all-gathered): the shard of the 1 M x 512 store with its share of the planted
rows, all N x 1024 queries.  Prints the scan time, the search time and the certificate's statistics for that per-rank shape.
usage: python tools/rehearse_rank.py --world 8 [--rank 0]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import shard_bounds

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--plant", choices=["rank0", "all"], default="rank0",
                help="rank0 (bench.py since round 5): the store holds near-duplicates of rank 0's 1024 clips only -- the same store for every N; "
                     "all (rounds 3-4): of every rank's clips -- planted rows, and with them every rank's candidates, grow with N")
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
B, S, N, D, K = 1024, 64000, 1_000_000, 512, 10
cfg = R.Config()
cfg.update(device=dev, tpp_levels=[1], tpp_pooling_type="max", feature_dim=D, vector_db_index_type="IP")
fe = R.MelProjectionFeatureExtractor(cfg)
embs = []
wave = torch.empty(B * S, device=dev)
offs = np.arange(B + 1, dtype=np.int64) * S
for r in range(a.world):
    _lib.check(lib.radad_synth_audio(wave.data_ptr(), r * B, B, S, 1234, 0, _lib.stream_ptr(dev)))
    embs.append(fe.embed_clips(wave, offs).clone())
all_emb = torch.cat(embs)
Q = a.world * B
lo, hi = shard_bounds(N, a.world, a.rank)
rows = torch.empty((hi - lo, D), device=dev)
_lib.check(lib.radad_synth_rows(rows.data_ptr(), lo, hi - lo, D, 4321, 0, _lib.stream_ptr(dev)))
P = Q if a.plant == "all" else B                      # clips that have planted near-duplicates in the store
noise = torch.empty((2 * P, D), device=dev)
_lib.check(lib.radad_synth_rows(noise.data_ptr(), 0, 2 * P, D, 99, 0, _lib.stream_ptr(dev)))
jj = torch.arange(P, device=dev)
pemb = all_emb[:P]
scale = pemb.norm(dim=1, keepdim=True) / (D ** 0.5)
for c, eps in ((0, 0.05), (1, 0.10)):
    g = (jj * 977 + c * 350003 + 17) % N
    mine = (g >= lo) & (g < hi)
    rows[g[mine] - lo] = pemb[mine] + eps * scale[mine] * noise[c * P:(c + 1) * P][mine]
idx = R.HipFlatIndex(D, _lib.METRIC_COSINE, 0, id_base=lo)
idx.add_device(rows)
# the bound the OTHER shards would contribute to the all-reduce: their search_begin over their rows (built one after the other here)
ap_bound = a.world > 1
glb = None
if ap_bound:
    for r in range(a.world):
        if r == a.rank:
            continue
        l2_, h2_ = shard_bounds(N, a.world, r)
        orow = torch.empty((h2_ - l2_, D), device=dev)
        _lib.check(lib.radad_synth_rows(orow.data_ptr(), l2_, h2_ - l2_, D, 4321, 0, _lib.stream_ptr(dev)))
        for c, eps in ((0, 0.05), (1, 0.10)):
            g = (jj * 977 + c * 350003 + 17) % N
            mine = (g >= l2_) & (g < h2_)
            orow[g[mine] - l2_] = pemb[mine] + eps * scale[mine] * noise[c * P:(c + 1) * P][mine]
        other = R.HipFlatIndex(D, _lib.METRIC_COSINE, 0, id_base=l2_)
        other.add_device(orow)
        lb = other.search_begin(all_emb, K)
        other.search_finish(None)
        glb = lb if glb is None else torch.cat([glb, lb], 1)      # [Q, (G - 1) k] lower bounds from the other shards
        del other, orow
idx.profile(True)


def one(bound):
    lb = idx.search_begin(all_emb, K)
    t_a = torch.cuda.Event(enable_timing=True); t_b = torch.cuda.Event(enable_timing=True)
    t_a.record()
    g = None
    if bound and glb is not None:
        allb = torch.cat([glb, lb], 1)
        g = R.HipFlatIndex.global_bound(allb.view(allb.shape[0], a.world, K).permute(1, 0, 2).contiguous(), K)
    idx.search_finish(g, return_f64=True)
    t_b.record()
    return t_a, t_b


for bound in ((False, True) if ap_bound else (False,)):
    for _ in range(2):
        one(bound)
    idx.profile_read()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    evs = []
    t0.record()
    for _ in range(a.reps):
        evs.append(one(bound))
    t1.record(); torch.cuda.synchronize()
    ms = idx.profile_read()
    la = idx.last_launch()
    print(json.dumps({"world": a.world, "rank": a.rank, "plant": a.plant, "rows": hi - lo, "queries": Q, "global_bound": bound,
                      "scan_ms": round(float(np.mean(ms)) * la["scan_launches"], 4), "rerank_ms": round(float(np.mean([x.elapsed_time(y) for x, y in evs])), 4),
                      "search_ms": round(t0.elapsed_time(t1) / a.reps, 4),
                      "candidates_per_query": round(la["certificate"]["candidates_rescored"] / Q, 1), "launch": la}))
