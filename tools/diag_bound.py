#!/usr/bin/env python3
"""diagnostic: what the cross-shard bound does at the benchmark's per-rank shape (one GPU, all shards built one after the other)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import shard_bounds
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0"); lib = _lib.load()
B, S, N, D, K = 1024, 64000, 1_000_000, 512, 10
cfg = R.Config(); cfg.update(device=dev, tpp_levels=[1], tpp_pooling_type="max", feature_dim=D, vector_db_index_type="IP")
fe = R.MelProjectionFeatureExtractor(cfg)
wave = torch.empty(B * S, device=dev); offs = np.arange(B + 1, dtype=np.int64) * S
embs = []
for r in range(G):
    _lib.check(lib.radad_synth_audio(wave.data_ptr(), r * B, B, S, 1234, 0, _lib.stream_ptr(dev)))
    embs.append(fe.embed_clips(wave, offs).clone())
q = torch.cat(embs); Q = G * B
qn = q / q.norm(dim=1, keepdim=True)
print("cos between embeddings: mean %.6f min %.6f" % (float((qn[:64] @ qn[:2048].T).mean()), float((qn[:64] @ qn[:2048].T).min())))
noise = torch.empty((2 * Q, D), device=dev); _lib.check(lib.radad_synth_rows(noise.data_ptr(), 0, 2 * Q, D, 99, 0, _lib.stream_ptr(dev)))
jj = torch.arange(Q, device=dev); scale = q.norm(dim=1, keepdim=True) / (D ** 0.5)
lbs, idxs, sims = [], [], []
for r in range(G):
    lo, hi = shard_bounds(N, G, r)
    rows = torch.empty((hi - lo, D), device=dev); _lib.check(lib.radad_synth_rows(rows.data_ptr(), lo, hi - lo, D, 4321, 0, _lib.stream_ptr(dev)))
    for c, e in ((0, 0.05), (1, 0.10)):
        g = (jj * 977 + c * 350003 + 17) % N; mine = (g >= lo) & (g < hi)
        rows[g[mine] - lo] = q[mine] + e * scale[mine] * noise[c * Q:(c + 1) * Q][mine]
    idx = R.HipFlatIndex(D, _lib.METRIC_COSINE, 0, id_base=lo); idx.add_device(rows)
    lb = idx.search_begin(q, K); idx.search_finish(None)
    lbs.append(lb.clone())
    rn = rows / rows.norm(dim=1, keepdim=True)
    s = (qn[:8] @ rn.T)                     # exact-ish sims of 8 queries against this shard
    sims.append(s.sort(dim=1, descending=True).values[:, :400].cpu())
    if r == 0:
        idx0 = idx; rows0 = rows
    else:
        del idx
allb = torch.stack(lbs)                      # [G, Q, K]
glob = torch.topk(allb.permute(1, 0, 2).reshape(Q, G * K), K, dim=1).values[:, K - 1]
loc = lbs[0].min(dim=1).values
print("query 0: shard-0 top-10 lower bounds", [round(float(x), 5) for x in lbs[0][0].sort(descending=True).values])
print("query 0: global bound %.5f, local k-th bound %.5f" % (float(glob[0]), float(loc[0])))
print("query 0: shard 0 sims sorted (top 30):", [round(float(x), 5) for x in sims[0][0][:30]])
for thr_name, thr in (("local", loc), ("global", glob)):
    cnt = [(int((sims[0][i] >= float(thr[i]) - 5e-4).sum())) for i in range(8)]
    print(thr_name, "rows of shard 0 above (bound - 5e-4) for 8 queries:", cnt)
lb0 = idx0.search_begin(q, K); idx0.search_finish(glob, return_f64=True); print("with bound:", idx0.last_launch()["certificate"])
lb0 = idx0.search_begin(q, K); idx0.search_finish(None, return_f64=True); print("without   :", idx0.last_launch()["certificate"])
