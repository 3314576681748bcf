import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load()
B, N, D, K = 1024, 1_000_000, 512, 10
cfg = R.Config(); cfg.update(device=dev, feature_dim=512, tpp_levels=[1], tpp_pooling_type="max", segment_length=2.0, segment_overlap=0.5)
fe = R.MelProjectionFeatureExtractor(cfg)
wave = torch.empty(B * 64000, device=dev)
_lib.check(lib.radad_synth_audio(wave.data_ptr(), 0, B, 64000, 1234, 0, _lib.stream_ptr(dev)))
offs = np.arange(B + 1, dtype=np.int64) * 64000
idx = R.HipFlatIndex(D, _lib.METRIC_COSINE, 0)
for r0 in range(0, N, 1 << 18):
    n = min(1 << 18, N - r0); rows = torch.empty((n, D), device=dev)
    _lib.check(lib.radad_synth_rows(rows.data_ptr(), r0, n, D, 4321, 0, _lib.stream_ptr(dev))); idx.add_device(rows)
def run(steps):
    for _ in range(steps):
        e = fe.embed_clips(wave, offs); idx.search_device(e, K)
for name, pe, pi in (("no events", False, False), ("scan events", False, True), ("all events", True, True), ("no events", False, False), ("all events", True, True)):
    fe.profile(pe); idx.profile(pi)
    run(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    if pe: fe.profile_read()
    if pi: idx.profile_read()
    print(f"{name:12s} {dt*1e3:.4f} ms/step {B/dt/1e3:.1f} k")
