#!/usr/bin/env python3
"""(test infrastructure, run by hand: python tests/fuzz_embed.py [seed] [cases])
Randomised parity sweep of radad_embed_forward against the numpy float64 oracle: random segment lengths / overlaps, clip
lengths (shorter than a segment, exact multiples, tails), feature dims, pyramid levels, pooling modes, normalisation on/off,
amplitudes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from oracle import radad_oracle as O, synth

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
bad = 0
for case in range(ncase):
    seg_s = float(rng.choice([0.03, 0.1, 0.5, 1.0, 1.37, 2.0]))
    seg_s = round(seg_s * 100) / 100
    L = int(seg_s * 16000) // 160 * 160
    seg_s = L / 16000
    overlap = float(rng.choice([0.0, 0.25, 0.5, 0.75]))
    F = int(rng.choice([32, 64, 96, 256, 512]))
    levels = [[1], [1, 2], [1, 2, 4], [1, 3, 5], [2]][rng.integers(5)]
    mode = ["max", "avg"][rng.integers(2)]
    norm = bool(rng.integers(2))
    gain = float(rng.choice([1.0, 1e-4, 3e3]))
    cfg = R.Config()
    cfg.update(device=dev, feature_dim=F, tpp_levels=levels, tpp_pooling_type=mode, segment_length=seg_s, segment_overlap=overlap,
               melproj_normalize=norm, melproj_seed=100 + case)
    fe = R.MelProjectionFeatureExtractor(cfg)
    nclip = int(rng.integers(1, 6))
    lens = [int(x) for x in rng.choice([1, L // 3 + 1, L - 1, L, L + 1, 2 * L + 7, 3 * fe.hop_length + L, 50000], size=nclip)]
    wav = synth.audio(0, nclip, max(lens), 7000 + case) * np.float32(gain)
    clips = [wav[i, :n] for i, n in enumerate(lens)]
    offs = np.concatenate([[0], np.cumsum(lens)])
    emb = fe.embed_clips(torch.from_numpy(np.concatenate(clips)).to(dev), offs).cpu().numpy()
    ref = O.embed_clips(clips, fe.segment_length, fe.hop_length, fe.proj_w, fe.proj_b, tuple(levels), mode, normalize=norm)
    err = float(np.abs(emb - ref).max())
    ok = err < 1e-4 and emb.shape == ref.shape
    print(json.dumps(dict(case=case, L=L, hop=fe.hop_length, F=F, levels=levels, mode=mode, norm=norm, gain=gain, lens=lens, err=err, ok=ok)), flush=True)
    bad += 0 if ok else 1
print("FUZZ", "FAILED" if bad else "ok", bad, "bad of", ncase)
sys.exit(1 if bad else 0)
