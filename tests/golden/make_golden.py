"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own modules
(/root/reference: segmenter.py, pooling.py, projection.py, radad_model.py) and the HuggingFace front-ends the
reference's extractors call (default constructors, no from_pretrained / no network).

Run here (the container that has /root/reference); the GPU box never sees the reference:
    python tests/golden/make_golden.py
Only DATA is written (inputs by seed or by value, outputs by value) -- no reference source.

The reference's config.py imports torchaudio / faiss / librosa at module top (config.py:3,14,16) although the
modules imported here never use them; those three names are registered as empty modules so the import
resolves (this is the procedure SURVEY.md section 8c records).  faiss itself is NOT emulated: the kNN has no
golden vectors from the reference (its arithmetic lives in faiss, absent here) -- see oracle/__init__.py.
"""
import importlib.machinery
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    import torch  # noqa: F401
    import transformers  # noqa: F401
    for name in ("torchaudio", "faiss", "librosa"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
    sys.path.insert(0, REF)
    import config as ref_config
    import segmenter as ref_segmenter
    import pooling as ref_pooling
    import projection as ref_projection
    import radad_model as ref_radad
    return ref_config, ref_segmenter, ref_pooling, ref_projection, ref_radad


def main():
    import torch
    ref_config, ref_segmenter, ref_pooling, ref_projection, ref_radad = import_reference()
    cfg = ref_config.Config()
    cfg.device = torch.device("cpu")
    rng = np.random.default_rng(7)

    # ---- a1 segmenter.py:15-49 ---------------------------------------------------------------------------
    seg = ref_segmenter.AudioSegmenter(cfg)
    lengths = [100, 16000, 31999, 32000, 32001, 47999, 48000, 64000, 70001, 80000]
    g = {"segment_length": seg.segment_length, "hop_length": seg.hop_length, "lengths": np.asarray(lengths)}
    for n in lengths:
        audio = (np.arange(n, dtype=np.float32) % 977) / np.float32(977.0) - np.float32(0.5)
        segs = seg.segment_audio(audio)
        g[f"n{n}_count"] = np.asarray(len(segs))
        g[f"n{n}_dtype"] = np.asarray(str(segs[-1].dtype))
        g[f"n{n}_first8"] = np.stack([np.asarray(s[:8], np.float64) for s in segs])
        g[f"n{n}_last8"] = np.stack([np.asarray(s[-8:], np.float64) for s in segs])
        g[f"n{n}_sum"] = np.asarray([np.asarray(s, np.float64).sum() for s in segs])
    np.savez_compressed(os.path.join(OUT, "segmenter.npz"), **g)

    # ---- a3 pooling.py:88-122 and a4 pipeline.py:411 --------------------------------------------------------
    g = {}
    cases = []
    for mode in ("max", "avg"):
        for levels in ([1, 2, 4], [1], [1, 3, 5]):
            for (T, F) in ((1, 8), (3, 8), (4, 8), (7, 8), (99, 32), (200, 64), (1500, 16)):
                cfg.tpp_levels, cfg.tpp_pooling_type, cfg.feature_dim = levels, mode, F
                tpp = ref_pooling.TemporalPyramidPooling(cfg)
                x = rng.standard_normal((T, F)).astype(np.float32)
                y = tpp.pool_features(torch.from_numpy(x)).numpy()
                key = f"{mode}_{'-'.join(map(str, levels))}_{T}x{F}"
                cases.append(key)
                g[key + "_x"], g[key + "_y"] = x, y
                assert tpp.get_output_dim() == y.shape[0]
    g["cases"] = np.asarray(cases)
    # segment mean: torch.mean(torch.stack(seg_pooled), dim=0)
    for S in (1, 2, 3):
        v = rng.standard_normal((S, 56)).astype(np.float32)
        g[f"segmean{S}_x"] = v
        g[f"segmean{S}_y"] = torch.mean(torch.stack([torch.from_numpy(r) for r in v]), dim=0).numpy()
    np.savez_compressed(os.path.join(OUT, "pooling.npz"), **g)

    # ---- a8 projection.py:68-106 and radad_model.py:32-41 ---------------------------------------------------
    # Weights are NOT stored: both sides regenerate them with oracle.synth.fill_state_dict(shapes, seed).
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import synth
    g = {}
    cfg.tpp_levels, cfg.tpp_pooling_type = [1, 2, 4], "max"
    for D, B, seed in ((512, 3, 5100), (3584, 2, 5200)):
        layer = ref_projection.ProjectionLayer(cfg, D).eval()
        sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in layer.state_dict().items()}, seed)
        layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        x = torch.from_numpy(synth.rows(0, B * 5, D, seed + 99).reshape(B, 5, D))
        with torch.no_grad():
            y = layer(x)
        g[f"proj{D}_seed"] = np.asarray(seed)
        g[f"proj{D}_names"] = np.asarray(list(sd.keys()))
        g[f"proj{D}_x"], g[f"proj{D}_y"] = x.numpy(), y.numpy()
    D, seed = 512, 5300
    model = ref_radad.RADADModel(cfg, D).eval()
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = torch.from_numpy(synth.rows(0, 15, D, seed + 98).reshape(3, 5, D))
    t = torch.from_numpy(synth.rows(0, 3, D, seed + 97))
    with torch.no_grad():
        proj = model.projection_layer(x)
        fused = model.fuse(torch.cat([t, proj], dim=1))
        logits = model(x, t)
    g["radad_seed"] = np.asarray(seed)
    g["radad_names"] = np.asarray(list(sd.keys()))
    g["radad_shapes"] = np.asarray([",".join(map(str, v.shape)) for v in sd.values()])
    g["radad_x"], g["radad_t"], g["radad_proj"], g["radad_fused"], g["radad_logits"] = \
        x.numpy(), t.numpy(), proj.numpy(), fused.numpy(), np.atleast_1d(logits.numpy())
    np.savez_compressed(os.path.join(OUT, "projection.npz"), **g)

    # ---- a2 front-ends: the HF extractors feature_extractor.py:25-30 / :94-97 call ------------------------------
    from transformers import Wav2Vec2FeatureExtractor, WhisperFeatureExtractor
    import transformers
    g = {"transformers_version": np.asarray(transformers.__version__)}
    t_ = np.arange(32000) / 16000.0
    seg_a = (0.1 * rng.standard_normal(32000) + 0.3 * np.sin(2 * np.pi * 440.0 * t_)).astype(np.float32)
    seg_b = (0.05 * rng.standard_normal(32000) + 0.2 * np.sign(np.sin(2 * np.pi * 97.0 * t_)) * t_).astype(np.float32)
    segs = np.stack([seg_a, seg_b])
    g["segments"] = segs
    w2v = Wav2Vec2FeatureExtractor()      # do_normalize=True, return_attention_mask=False
    norm = w2v([s for s in segs], sampling_rate=16000, return_tensors="np", padding=True).input_values
    g["w2v_normalized"] = np.asarray(norm, np.float32)
    wfe = WhisperFeatureExtractor()       # n_fft 400, hop 160, 80 mels, 30 s padding
    g["mel_filters"] = np.asarray(wfe.mel_filters, np.float64)              # [201, 80]
    full = np.concatenate([wfe(s, sampling_rate=16000, return_tensors="np").input_features for s in segs])  # [2,80,3000]
    g["whisper_padded_first208"] = np.asarray(full[:, :, :208], np.float32)  # frames 0..207 (rest is constant)
    g["whisper_padded_tail"] = np.asarray(full[:, :, -1], np.float32)        # the constant silence frame
    g["whisper_padded_const_ok"] = np.asarray(bool(np.all(full[:, :, 208:] == full[:, :, -1:])))
    # spectrogram of the 2 s segment itself (no 30 s padding): same HF code path, padding disabled
    self_mode = np.concatenate([wfe(s, sampling_rate=16000, return_tensors="np", padding=False, truncation=False).input_features
                                for s in segs])
    g["whisper_self"] = np.asarray(self_mode, np.float32)                    # [2, 80, 200]
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), **g)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
