#!/usr/bin/env python3
"""Experiment (not product): the benchmark's step with the embedding of batch i + 1 on one stream and the search of batch i on
another, against the plain sequential loop -- does filling the search's small kernels with log-mel workgroups buy anything?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
B, N, D, K = 1024, 1_000_000, 512, 10
cfg = R.Config()
cfg.update(device=dev, feature_dim=512, tpp_levels=[1], tpp_pooling_type="max", segment_length=2.0, segment_overlap=0.5)
fe = R.MelProjectionFeatureExtractor(cfg)
wave = torch.empty(B * 64000, device=dev)
_lib.check(lib.radad_synth_audio(wave.data_ptr(), 0, B, 64000, 1234, 0, _lib.stream_ptr(dev)))
offs = np.arange(B + 1, dtype=np.int64) * 64000
idx = R.HipFlatIndex(D, _lib.METRIC_COSINE, 0)
for r0 in range(0, N, 1 << 18):
    n = min(1 << 18, N - r0)
    rows = torch.empty((n, D), device=dev)
    _lib.check(lib.radad_synth_rows(rows.data_ptr(), r0, n, D, 4321, 0, _lib.stream_ptr(dev)))
    idx.add_device(rows)


def sequential(steps):
    for _ in range(steps):
        e = fe.embed_clips(wave, offs)
        idx.search_device(e, K)


def overlapped(steps, prio):
    sE = torch.cuda.Stream(priority=0)
    sS = torch.cuda.Stream(priority=-1 if prio else 0)
    cur = torch.cuda.current_stream()
    sE.wait_stream(cur); sS.wait_stream(cur)
    for _ in range(steps):
        with torch.cuda.stream(sE):
            e = fe.embed_clips(wave, offs)
            ev = torch.cuda.Event(); ev.record(sE)
        with torch.cuda.stream(sS):
            sS.wait_event(ev)
            e.record_stream(sS)
            idx.search_device(e, K)
    cur.wait_stream(sE); cur.wait_stream(sS)


for name, fn in (("sequential", lambda s: sequential(s)), ("two streams", lambda s: overlapped(s, False)),
                 ("two streams, search high priority", lambda s: overlapped(s, True)), ("sequential", lambda s: sequential(s))):
    fn(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(40)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"{name:36s} {dt * 1e3:.4f} ms per step  {B / dt / 1e3:.1f} k clips/s")


# ---- the front-end of one sub-batch beside the projection of another -- two extractor handles (each owns its log-mel
# scratch), half a batch each, on two streams: k_logmel_fft_clip (vector ALU) of one half could run beside k_proj_pool2 (matrix pipe) of
# the other IF both fit a CU at once.  They do not: 237 + 2 x 242 VGPRs per SIMD lane-row against 512; this measures it.
fe2 = R.MelProjectionFeatureExtractor(cfg)
H = B // 2
w1, w2 = wave[:H * 64000], wave[H * 64000:]
o_h = np.arange(H + 1, dtype=np.int64) * 64000


def embed_whole(steps):
    for _ in range(steps):
        fe.embed_clips(wave, offs)


def embed_halves_one_stream(steps):
    for _ in range(steps):
        fe.embed_clips(w1, o_h); fe2.embed_clips(w2, o_h)


def embed_halves_two_streams(steps):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for _ in range(steps):
        with torch.cuda.stream(s1):
            fe.embed_clips(w1, o_h)
        with torch.cuda.stream(s2):
            fe2.embed_clips(w2, o_h)
    cur.wait_stream(s1); cur.wait_stream(s2)


for name, fn in (("embed, whole batch", embed_whole), ("embed, two halves, one stream", embed_halves_one_stream),
                 ("embed, two halves, two streams", embed_halves_two_streams), ("embed, whole batch", embed_whole)):
    fn(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(40)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"{name:36s} {dt * 1e3:.4f} ms per {B} clips")
