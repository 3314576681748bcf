#!/usr/bin/env python3
"""Embedding-kernel timing without the store: 1024 clips x 4 s through radad_embed_forward, HIP-event kernel times."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load()
B, S = 1024, 64000
cfg = R.Config(); cfg.update(device=dev, feature_dim=512, tpp_levels=[1])
fe = R.build_feature_extractor(cfg)
wave = torch.empty(B * S, device=dev)
_lib.check(lib.radad_synth_audio(wave.data_ptr(), 0, B, S, 1234, 0, _lib.stream_ptr(dev)))
offs = np.arange(B + 1, dtype=np.int64) * S
for _ in range(3): fe.embed_clips(wave, offs)
fe.profile(True)
for _ in range(10): e = fe.embed_clips(wave, offs)
torch.cuda.synchronize()
lm, pp = fe.profile_read()
print(json.dumps({"k_logmel_ms": round(float(np.mean(lm)), 4), "k_proj_pool_ms": round(float(np.mean(pp)), 4), "checksum": float(e.double().sum())}))
