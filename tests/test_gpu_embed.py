"""GPU parity: csrc/embed.hip stage by stage and fused, against the oracle and the golden vectors.
Tolerance: 1e-4 absolute on embeddings / log-mel values (north_star), bit-exact for max pooling."""
import os

import numpy as np
import pytest

from oracle import radad_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu


def _fe(gpu, **kw):
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    cfg = R.Config()
    cfg.update(device=gpu, **kw)
    return R.MelProjectionFeatureExtractor(cfg), cfg


@pytest.fixture(scope="module")
def g_fe(golden_dir):
    return np.load(os.path.join(golden_dir, "frontend.npz"))


def test_normalize_matches_hf_golden(gpu, g_fe):
    fe, _ = _fe(gpu, feature_dim=32, tpp_levels=[1])
    y = fe.normalize_segments(list(g_fe["segments"])).cpu().numpy()
    np.testing.assert_allclose(y, g_fe["w2v_normalized"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(y, O.zero_mean_unit_var(g_fe["segments"]), rtol=0, atol=2e-5)
    # a short clip: the segmenter's zero padding is part of the statistics (segmenter.py:34-37)
    short = g_fe["segments"][0][:12345]
    y = fe.normalize_segments([short]).cpu().numpy()[0]
    ref = O.zero_mean_unit_var(np.concatenate([short, np.zeros(32000 - 12345)]))
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-5)


def test_mel_filter_bank_matches_hf_golden(g_fe):
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.feature_extractor import mel_filter_bank_slaney
    np.testing.assert_allclose(mel_filter_bank_slaney(), g_fe["mel_filters"], rtol=1e-12, atol=1e-15)


def test_logmel_self_mode(gpu, g_fe):
    fe, _ = _fe(gpu, feature_dim=32, tpp_levels=[1], melproj_normalize=False)
    lm = fe.log_mel(list(g_fe["segments"])).cpu().numpy()            # [2, 200, 80]
    assert lm.shape == (2, 200, 80)
    for s in range(2):
        np.testing.assert_allclose(lm[s], O.log_mel(g_fe["segments"][s]), rtol=0, atol=1e-4)
        np.testing.assert_allclose(lm[s].T, g_fe["whisper_self"][s], rtol=0, atol=3e-4)      # HF's own fp32 FFT


def test_logmel_padded_mode(gpu, g_fe):
    fe, _ = _fe(gpu, feature_dim=32, tpp_levels=[1], melproj_normalize=False, melproj_padded_samples=480000)
    assert fe.num_frames == 3000
    lm = fe.log_mel(list(g_fe["segments"])).cpu().numpy()            # [2, 3000, 80]
    for s in range(2):
        np.testing.assert_allclose(lm[s], O.log_mel(g_fe["segments"][s], padded_samples=480000), rtol=0, atol=1e-4)
        np.testing.assert_allclose(lm[s, :208].T, g_fe["whisper_padded_first208"][s], rtol=0, atol=3e-4)
        np.testing.assert_allclose(lm[s, -1], g_fe["whisper_padded_tail"][s], rtol=0, atol=1e-5)


def test_frame_features_protocol(gpu, g_fe):
    """extract_features(list of segments) -> list of [T, F] tensors on config.device (feature_extractor.py:21-52)"""
    fe, cfg = _fe(gpu, feature_dim=96, tpp_levels=[1, 2, 4])
    segs = [g_fe["segments"][0], g_fe["segments"][1][:20000]]
    feats = fe.extract_features(segs)
    assert len(feats) == 2 and feats[0].shape == (200, 96) and feats[0].is_cuda and fe.feature_dim == 96
    for s, seg in enumerate(segs):
        x = O.zero_mean_unit_var(np.concatenate([seg, np.zeros(32000 - len(seg))]))
        ref = O.frame_projection(O.log_mel(x), fe.proj_w, fe.proj_b)
        np.testing.assert_allclose(feats[s].cpu().numpy(), ref, rtol=0, atol=1e-4)
    assert fe.extract_features([]) == []


@pytest.mark.parametrize("gain", [1e-6, 1.0, 3e4])
def test_logmel_split_scales(gpu, g_fe, gain):
    """k_logmel_h carries every fp32 sample as two f16 numbers under a per-segment power-of-two scale: amplitudes from 1e-6 to
    int16-range, an all-zero segment and a segment with one huge spike keep the 1e-4 bar (log10 moves by log10(gain^2))."""
    fe, _ = _fe(gpu, feature_dim=32, tpp_levels=[1], melproj_normalize=False)
    seg = (g_fe["segments"][0] * np.float32(gain)).astype(np.float32)
    spike = seg.copy()
    spike[12345] = np.float32(500.0 * gain)                       # dominates the scale: the rest lives in the lo halves
    zero = np.zeros_like(seg)
    lm = fe.log_mel([seg, spike, zero]).cpu().numpy()
    np.testing.assert_allclose(lm[0], O.log_mel(seg), rtol=0, atol=1e-4)
    # frames the spike's window does not reach are unchanged apart from the segment-wide max - 8 clamp
    far = np.ones(200, bool)
    far[12345 // 160 - 2: 12345 // 160 + 4] = False
    ref_spike = O.log_mel(spike)
    np.testing.assert_allclose(lm[1][~far], ref_spike[~far], rtol=0, atol=1e-4)
    np.testing.assert_allclose(lm[1][far], ref_spike[far], rtol=0, atol=2e-3)   # 1e-4 relative to the spike's 5 decades
    np.testing.assert_allclose(lm[2], O.log_mel(zero), rtol=0, atol=1e-6)


def test_projection_column_scales(gpu, g_fe, tmp_path):
    """k_proj_pool splits W per feature column: columns of very different magnitude (and a zero column) keep 1e-4 relative to
    their own scale"""
    import torch
    F = 64
    w = synth.rows(0, 80, F, 41) * np.exp2((np.arange(F) % 21) - 10).astype(np.float32)[None, :]
    w[:, 5] = 0
    b = synth.rows(0, 1, F, 42)[0]
    path = str(tmp_path / "w.npz")
    np.savez(path, w=w, b=b)
    fe, _ = _fe(gpu, feature_dim=F, tpp_levels=[1], melproj_weights_path=path)
    feats = fe.extract_features([g_fe["segments"][0]])[0].cpu().numpy()
    x = O.zero_mean_unit_var(g_fe["segments"][0])
    ref = O.frame_projection(O.log_mel(x), w, b)
    scale = np.abs(w).max(axis=0) + 1e-30
    np.testing.assert_allclose((feats - b) / scale, (ref - b) / scale, rtol=0, atol=2e-4)
    np.testing.assert_allclose(feats[:, 5], b[5], rtol=0, atol=0)


def test_log_mel_kernels_agree(gpu):
    """the four log-mel kernels -- radix FFT on the vector ALU (default), DFT-as-GEMM on the f16 matrix pipe with and without
    shared frames, folded DFT on the fp32 matrix pipe -- chosen through the extractor's knobs (radad_embed_create_ex flags): all
    within the parity bar of the float64 oracle and of each other"""
    import torch
    wav = synth.audio(0, 4, 64000, 77)
    offs = np.arange(5) * 64000
    out, kinds = [], []
    for kw in (dict(), dict(melproj_logmel_fft=False), dict(melproj_share_frames=False), dict(melproj_logmel_f32=True)):
        fe, _ = _fe(gpu, feature_dim=128, tpp_levels=[1, 2], **kw)
        out.append(fe.embed_clips(torch.from_numpy(wav.reshape(-1)).to(gpu), offs).cpu().numpy())
        kinds.append(fe.last_logmel_kind())
    assert kinds == ["clip_frames_fft", "clip_frames", "per_segment", "per_segment"]
    ref = O.embed_clips(list(wav), 32000, 16000, fe.proj_w, fe.proj_b, (1, 2), "max")
    for o in out:
        np.testing.assert_allclose(o, ref, rtol=0, atol=1e-4)
        np.testing.assert_allclose(o, out[0], rtol=0, atol=5e-5)


def test_tpp_matches_reference_golden(gpu, golden_dir):
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    g = np.load(os.path.join(golden_dir, "pooling.npz"))
    cfg = R.Config()
    cfg.update(device=gpu)
    for key in g["cases"]:
        key = str(key)
        mode, levels, _ = key.split("_")
        cfg.tpp_levels, cfg.tpp_pooling_type = [int(v) for v in levels.split("-")], mode
        cfg.feature_dim = g[key + "_x"].shape[1]
        tpp = R.TemporalPyramidPooling(cfg)
        y = tpp.pool_features(torch.from_numpy(g[key + "_x"])).cpu().numpy()      # CPU tensor in -> moved, as the reference does
        assert y.shape[0] == tpp.get_output_dim()
        if mode == "max":
            np.testing.assert_array_equal(y, g[key + "_y"])
        else:
            np.testing.assert_allclose(y, g[key + "_y"], rtol=0, atol=2e-6)
    # batch form == per-item form (pooling.py:105-117)
    cfg.tpp_levels, cfg.tpp_pooling_type, cfg.feature_dim = [1, 2, 4], "max", 32
    tpp = R.TemporalPyramidPooling(cfg)
    xs = [torch.from_numpy(g["max_1-2-4_99x32_x"]).to(gpu), torch.from_numpy(g["avg_1-2-4_99x32_x"][:40]).to(gpu)]
    yb = tpp.pool_features_batch(xs)
    for i, x in enumerate(xs):
        np.testing.assert_array_equal(yb[i].cpu().numpy(), tpp.pool_features(x).cpu().numpy())
    cfg.tpp_pooling_type = "median"
    with pytest.raises(ValueError):
        R.TemporalPyramidPooling(cfg).pool_features(xs[0])


@pytest.mark.parametrize("levels,mode,F", [([1], "max", 512), ([1, 2, 4], "max", 128), ([1, 2, 4], "avg", 64), ([1, 3], "avg", 32)])
def test_embed_clips_fixed_length(gpu, levels, mode, F):
    import torch
    fe, _ = _fe(gpu, feature_dim=F, tpp_levels=levels, tpp_pooling_type=mode)
    B, n = 6, 64000                                    # 4 s clips -> 3 segments each
    wav = synth.audio(0, B, n, 1234)
    emb = fe.embed_clips(torch.from_numpy(wav.reshape(-1)).to(gpu), np.arange(B + 1) * n).cpu().numpy()
    assert emb.shape == (B, sum(levels) * F) == (B, fe.output_dim)
    ref = O.embed_clips(list(wav), 32000, 16000, fe.proj_w, fe.proj_b, levels, mode)
    np.testing.assert_allclose(emb, ref, rtol=0, atol=1e-4)


@pytest.mark.parametrize("seg_s,overlap", [(1.0, 0.5), (0.5, 0.25), (0.03, 0.0), (1.5, 0.5)])
def test_embed_other_segment_lengths(gpu, seg_s, overlap):
    """segment lengths other than the reference's 2 s (16000, 8000, 480 and 24000 samples): fewer frame tiles than waves
    (idle waves leave early), one- and three-tile waves, short clips padded"""
    import torch
    fe, _ = _fe(gpu, feature_dim=64, tpp_levels=[1, 2], segment_length=seg_s, segment_overlap=overlap)
    L, hop = fe.segment_length, fe.hop_length
    lens = [L // 2, L, 3 * L + 17, 40000]
    wav = synth.audio(0, len(lens), max(lens), 2235)
    clips = [wav[i, :n] for i, n in enumerate(lens)]
    offs = np.concatenate([[0], np.cumsum(lens)])
    emb = fe.embed_clips(torch.from_numpy(np.concatenate(clips)).to(gpu), offs).cpu().numpy()
    ref = O.embed_clips(clips, L, hop, fe.proj_w, fe.proj_b, (1, 2), "max")
    np.testing.assert_allclose(emb, ref, rtol=0, atol=1e-4)


def test_embed_clips_ragged_lengths(gpu):
    """variable-length clips (config 3): short clip zero-padded, dropped tails, 1..6 segments"""
    import torch
    fe, _ = _fe(gpu, feature_dim=64, tpp_levels=[1, 2, 4])
    lens = [100, 31999, 32000, 48000, 70001, 112000, 64000]
    wav = synth.audio(0, len(lens), max(lens), 1235)
    clips = [wav[i, :n] for i, n in enumerate(lens)]
    offs = np.concatenate([[0], np.cumsum(lens)])
    emb = fe.embed_clips(torch.from_numpy(np.concatenate(clips)).to(gpu), offs).cpu().numpy()
    ref = O.embed_clips(clips, 32000, 16000, fe.proj_w, fe.proj_b, (1, 2, 4), "max")
    np.testing.assert_allclose(emb, ref, rtol=0, atol=1e-4)
    # same batch again (cached plan) and a different batch (plan rebuilt)
    emb2 = fe.embed_clips(torch.from_numpy(np.concatenate(clips)).to(gpu), offs).cpu().numpy()
    np.testing.assert_array_equal(emb, emb2)
    emb3 = fe.embed_clips(torch.from_numpy(np.concatenate(clips[::-1])).to(gpu), np.concatenate([[0], np.cumsum(lens[::-1])])).cpu().numpy()
    np.testing.assert_allclose(emb3, ref[::-1], rtol=0, atol=1e-4)
    assert fe.embed_clips(torch.zeros(0, device=gpu), [0]).shape == (0, fe.output_dim)


def test_synth_device_equals_host(gpu):
    """the stateless generators produce identical bits on host and device"""
    import ctypes as C
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    lib = _lib.load()
    r = torch.empty((300, 512), device=gpu)
    _lib.check(lib.radad_synth_rows(r.data_ptr(), 12345, 300, 512, 4321, gpu.index or 0, _lib.stream_ptr(gpu)))
    np.testing.assert_array_equal(r.cpu().numpy(), synth.rows(12345, 300, 512, 4321))
    a = torch.empty((5, 64000), device=gpu)
    _lib.check(lib.radad_synth_audio(a.data_ptr(), 7, 5, 64000, 1234, gpu.index or 0, _lib.stream_ptr(gpu)))
    np.testing.assert_array_equal(a.cpu().numpy(), synth.audio(7, 5, 64000, 1234))


def test_int16_pcm_input_is_the_loaders_float(gpu):
    """embed_clips on 16-bit PCM == embed_clips on sample / 32768 in float32 (what librosa.load / soundfile hand the reference),
    bit for bit: the conversion is exact, the rest is the same kernels; odd lengths and an unaligned start take the scalar tail"""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    fe, _ = _fe(gpu, feature_dim=128, tpp_levels=[1, 2])
    rng = np.random.default_rng(5)
    lens = [48000, 33001, 64000, 16007]
    pcm = torch.from_numpy(rng.integers(-32768, 32767, size=sum(lens), endpoint=True).astype(np.int16)).to(gpu)
    offs = np.concatenate([[0], np.cumsum(lens)])
    a = fe.embed_clips(pcm, offs)
    b = fe.embed_clips(pcm.float() / 32768.0, offs)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    lib = _lib.load()
    out = torch.empty(1001, device=gpu)
    _lib.check(lib.radad_pcm16_to_f32(pcm.data_ptr() + 2, out.data_ptr(), 1001, gpu.index or 0, _lib.stream_ptr(gpu)))      # unaligned source
    assert torch.equal(out, pcm[1:1002].float() / 32768.0)
