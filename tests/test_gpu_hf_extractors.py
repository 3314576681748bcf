"""The reference's three named extractors (feature_extractor.py:6-52, :54-115, :117-170; selector pipeline.py:54-65) with the
front-end in HIP: for a locally stored encoder (tiny, seeded weights written to tmp_path with save_pretrained -- the reference's
checkpoints are not obtainable offline) the features must equal what the reference's own recipe gives -- the HuggingFace processor on
the CPU feeding the same encoder."""
import numpy as np
import pytest

from oracle import synth

SEGS = 3


def _segments():
    return [synth.audio(i, 1, 32000, 6100)[0] * np.float32(1.0 + i) + np.float32(0.01 * i) for i in range(SEGS)]


def _tiny_w2v2(kind, path):
    import torch
    import transformers as T
    torch.manual_seed(11)
    kw = dict(hidden_size=32, num_hidden_layers=4, num_attention_heads=2, intermediate_size=64, conv_dim=(16,) * 7,
              num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, vocab_size=32)
    if kind == "wav2vec2":
        model = T.Wav2Vec2Model(T.Wav2Vec2Config(**kw))
    else:
        model = T.WavLMModel(T.WavLMConfig(**kw))
    model.eval().save_pretrained(path)
    T.Wav2Vec2FeatureExtractor(do_normalize=True, return_attention_mask=False).save_pretrained(path)
    return model


def _tiny_whisper(path):
    import torch
    import transformers as T
    torch.manual_seed(12)
    model = T.WhisperModel(T.WhisperConfig(d_model=32, encoder_layers=2, decoder_layers=1, encoder_attention_heads=2,
                                           decoder_attention_heads=2, encoder_ffn_dim=64, decoder_ffn_dim=64, vocab_size=64,
                                           num_mel_bins=80, max_source_positions=1500, max_target_positions=16,
                                           pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1))
    model.eval().save_pretrained(path)
    T.WhisperFeatureExtractor().save_pretrained(path)
    return model


def test_named_extractors_need_a_local_directory():
    """(CPU) nothing is ever downloaded: a hub NAME that is not a local directory is a clear error, for all three kinds"""
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    for kind in ("wav2vec2", "whisper", "wavlm"):
        cfg = R.Config()
        cfg.feature_extractor_type = kind
        with pytest.raises(FileNotFoundError, match="never downloads"):
            R.build_feature_extractor(cfg)
    cfg.feature_extractor_type = "hubert"
    with pytest.raises(ValueError, match="Unsupported feature_extractor_type"):
        R.build_feature_extractor(cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["wav2vec2", "wavlm"])
def test_wav2vec2_and_wavlm_adapters_match_the_hf_processor_path(gpu, tmp_path, kind):
    import torch
    import transformers as T
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    d = str(tmp_path / kind)
    _tiny_w2v2(kind, d)
    cfg = R.Config()
    cfg.update(device=gpu, feature_extractor_type=kind, **{f"{kind}_model_name": d})
    fe = R.build_feature_extractor(cfg)
    assert type(fe).__name__ == {"wav2vec2": "Wav2Vec2FeatureExtractor", "wavlm": "WavLMFeatureExtractor"}[kind] and fe.feature_dim == 32
    segs = _segments()
    feats = fe.extract_features(segs)
    assert len(feats) == SEGS and tuple(feats[0].shape) == (99, 32)
    # the reference's recipe: HF processor on the CPU (numpy), the same encoder
    proc = T.Wav2Vec2FeatureExtractor.from_pretrained(d)
    with torch.no_grad():
        if kind == "wav2vec2":      # feature_extractor.py:25-39
            x = proc(segs, sampling_rate=16000, return_tensors="pt", padding=True).input_values.to(gpu)
            hs = fe.model(x, output_hidden_states=True).hidden_states
            ref = torch.mean(torch.stack([hs[i] for i in cfg.wav2vec2_layers_to_use]), dim=0)
            ref = [r for r in ref]
            assert feats[0].is_cuda
        else:                       # feature_extractor.py:148-168
            ref = []
            for w in segs:
                x = proc(raw_speech=w, sampling_rate=16000, return_tensors="pt").input_values.to(gpu)
                ref.append(fe.model(x).last_hidden_state.squeeze(0).cpu())
            assert not feats[0].is_cuda
    # the front-end itself: what the encoder is handed (K1, feature_extraction_wav2vec2.py:78-97) against the HF processor's numpy
    x_hf = proc(segs, sampling_rate=16000, return_tensors="pt", padding=True).input_values
    x_hip = fe._inputs(segs).cpu()
    in_err = float((x_hip - x_hf).abs().max())
    assert in_err < 2e-5, in_err                       # (tests/test_gpu_embed.py's bar for the same kernel)
    # ... and through the encoder (measured on MI355X: input 4.8e-7, output 2.2e-6 / 2.7e-6 on features up to 3.2 / 3.5)
    err = max(float((a.float().cpu() - b.float().cpu()).abs().max()) for a, b in zip(feats, ref))
    scale = max(float(b.abs().max()) for b in ref)
    print(f"{kind}: encoder input max err {in_err:.2e}; encoder output max err {err:.3e} (features up to {scale:.2f})")
    assert err < 1e-4
    # and through the pipeline shell (pipeline.py:392-414): protocol-only extractor, HIP pooling, segment mean
    cfg.vector_db_path = str(tmp_path / "vdb")
    pipe = R.HotPathPipeline(cfg, feature_extractor=fe)

    class DS:
        def load_audio(self, path):
            return synth.audio(7, 1, 48000, 6101)[0]
    emb = pipe.process_audio_batch(["/a.wav"], DS())
    assert tuple(emb.shape) == (1, 7 * 32)


@pytest.mark.gpu
def test_whisper_adapter_matches_the_hf_processor_path(gpu, tmp_path):
    import torch
    import transformers as T
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    d = str(tmp_path / "whisper")
    _tiny_whisper(d)
    cfg = R.Config()
    cfg.update(device=gpu, feature_extractor_type="whisper", whisper_model_name=d)
    fe = R.build_feature_extractor(cfg)
    assert type(fe).__name__ == "WhisperFeatureExtractor" and fe.feature_dim == 32
    segs = _segments()[:2]
    feats = fe.extract_features(segs)
    assert len(feats) == 2 and tuple(feats[0].shape) == (1500, 32) and not feats[0].is_cuda
    proc = T.WhisperFeatureExtractor.from_pretrained(d)
    errs, mel_errs = [], []
    with torch.no_grad():
        for w, f in zip(segs, feats):                                     # feature_extractor.py:92-112
            x = proc(w, sampling_rate=16000, return_tensors="pt").input_features.to(gpu)        # [1, 80, 3000]
            ref = fe.model.encoder(x).last_hidden_state.squeeze(0).cpu()
            errs.append(float((f - ref).abs().max()))
            mel = fe._front.log_mel([w]).transpose(1, 2)
            mel_errs.append(float((mel - x).abs().max()))
    print(f"whisper: log-mel max err {max(mel_errs):.3e}; encoder output max err {max(errs):.3e}")
    assert max(mel_errs) < 3e-4          # the front-end bar of tests/test_gpu_embed.py::test_logmel_padded_mode (HF's fp32 FFT); measured 1.0e-6
    assert max(errs) < 1e-4              # measured 3.6e-7
