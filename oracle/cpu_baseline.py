"""The CPU baseline bench.py times: the same segment -> embed -> retrieve pipeline as oracle/radad_oracle.py, vectorised
in float32 on torch-CPU (FFT + BLAS, all host cores).  TEST / BENCH INFRASTRUCTURE ONLY: never imported by the product.

This is what a careful CPU deployment of the reference's path would run (the reference itself calls HF front-ends and
faiss-cpu, which are float32 FFT/BLAS code): SURVEY.md 8(d) asks for "C++/OpenMP or torch-CPU BLAS ... threads = all
physical cores".  The float64 oracle stays the CHECKER of the GPU result; this module is only ever the thing timed.

Stages (same citations as radad_oracle.py):
  segmenter.py:25-39              segment plan (fixed-length clips: a strided view; ragged: gather)
  feature_extraction_wav2vec2:95  zero-mean / unit-variance per segment, eps 1e-7
  feature_extraction_whisper:135+ reflect pad 200, hann(400) frames hop 160, rfft, |X|^2, drop last frame, mel 201->80,
                                  log10 clamp 1e-10, max-8, (x+4)/4
  frame projection + pooling.py:66-103 + pipeline.py:411
  vector_database.py:100-105,159-182  cosine = normalise + inner product; top-k by torch.topk on the float32 GEMM
"""
import os

import numpy as np


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _torch(threads):
    import torch
    torch.set_num_threads(threads)
    return torch


def embed_clips(clips, seg, hop, w, b, mel_filters, levels=(1,), mode="max", threads=None):
    """clips: list of 1-D float32 numpy arrays (or a [B, N] array).  -> float32 numpy [B, sum(levels)*F]."""
    threads = threads or len(os.sched_getaffinity(0))
    torch = _torch(threads)
    W = torch.from_numpy(np.ascontiguousarray(w, np.float32))
    bias = torch.from_numpy(np.ascontiguousarray(b, np.float32))
    fb = torch.from_numpy(np.ascontiguousarray(mel_filters, np.float32))          # [201, 80]
    win = torch.hann_window(400, periodic=True)
    segs, owner = [], []
    for ci, c in enumerate(clips):
        c = torch.from_numpy(np.ascontiguousarray(c, np.float32))
        n = c.numel()
        ns = max(1, (n - seg) // hop + 1)
        if n < seg:
            c = torch.cat([c, torch.zeros(seg - n)])
        segs.append(c.unfold(0, seg, hop)[:ns])
        owner += [ci] * ns
    x = torch.cat(segs)                                                            # [S, L]
    out_rows = []
    for s0 in range(0, x.shape[0], 512):                                           # bounded working set
        xs = x[s0:s0 + 512]
        xs = (xs - xs.mean(1, keepdim=True)) / torch.sqrt(xs.var(1, unbiased=False, keepdim=True) + 1e-7)
        spec = torch.stft(xs, 400, 160, window=win, center=True, pad_mode="reflect", return_complex=True)   # [S, 201, 1+L/160]
        power = (spec.real ** 2 + spec.imag ** 2)[:, :, :-1]                       # drop the last frame
        mel = torch.matmul(power.transpose(1, 2), fb)                              # [S, T, 80]
        logs = torch.log10(torch.clamp(mel, min=1e-10))
        logs = torch.maximum(logs, logs.amax(dim=(1, 2), keepdim=True) - 8.0)
        feats = torch.matmul((logs + 4.0) / 4.0, W) + bias                         # [S, T, F]
        T = feats.shape[1]
        pooled = []
        for l in levels:
            for i in range(l):
                lo, hi = (i * T) // l, -((-(i + 1) * T) // l)
                pooled.append(feats[:, lo:hi].amax(1) if mode == "max" else feats[:, lo:hi].mean(1))
        out_rows.append(torch.cat(pooled, 1))
    seg_vecs = torch.cat(out_rows)
    own = torch.tensor(owner)
    B = len(clips)
    acc = torch.zeros((B, seg_vecs.shape[1])).index_add_(0, own, seg_vecs)
    cnt = torch.zeros(B).index_add_(0, own, torch.ones(len(owner)))
    return (acc / cnt[:, None]).numpy()


def knn_cosine(db, q, k, threads=None, chunk=131072, db_is_normalised=False):
    """db [N, D], q [Q, D] float32 numpy -> (D f32 [Q,k], I i64 [Q,k]); float32 GEMM + topk per chunk, merged."""
    threads = threads or len(os.sched_getaffinity(0))
    torch = _torch(threads)
    qt = torch.from_numpy(np.ascontiguousarray(q, np.float32))
    qt = qt / (qt.norm(dim=1, keepdim=True) + 1e-12)
    best_d = torch.full((qt.shape[0], k), -float("inf"))
    best_i = torch.full((qt.shape[0], k), -1, dtype=torch.int64)
    for r0 in range(0, len(db), chunk):
        y = torch.from_numpy(db[r0:r0 + chunk])
        if not db_is_normalised:
            y = y / (y.norm(dim=1, keepdim=True) + 1e-12)
        s = qt @ y.T
        d, i = torch.topk(s, min(k, s.shape[1]), dim=1)
        cat_d, cat_i = torch.cat([best_d, d], 1), torch.cat([best_i, i + r0], 1)
        best_d, sel = torch.topk(cat_d, k, dim=1)
        best_i = torch.gather(cat_i, 1, sel)
    return best_d.numpy(), best_i.numpy()
