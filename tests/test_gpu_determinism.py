"""GPU: the embed -> search chain on fixed inputs gives BITWISE the same embeddings, ids and distances step after step -- nothing may
depend on timing (LDS-DMA landing, the hand-counted load queue of k_logmel_h, the slot buffers' atomics, which workgroup a CU
gets next).  tools/stress_determinism.py is the long form (900 steps at the benchmark's size: 0 mismatches)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric", ["IP", "L2"])
def test_chain_is_bitwise_repeatable(gpu, metric):
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    lib = _lib.load()
    B, S, N, D, K = 256, 64000, 200_000, 512, 10
    cfg = R.Config()
    cfg.update(device=gpu, tpp_levels=[1], tpp_pooling_type="max", feature_dim=D, vector_db_index_type=metric)
    fe = R.MelProjectionFeatureExtractor(cfg)
    wave = torch.empty(B * S, device=gpu)
    _lib.check(lib.radad_synth_audio(wave.data_ptr(), 0, B, S, 1234, 0, _lib.stream_ptr(gpu)))
    offs = np.arange(B + 1, dtype=np.int64) * S
    emb0 = fe.embed_clips(wave, offs).clone()
    rows = torch.empty((N, D), device=gpu)
    _lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, N, D, 4321, 0, _lib.stream_ptr(gpu)))
    jj = torch.arange(B, device=gpu)
    rows[(jj * 769 + 17) % N] = emb0 * 1.01
    vdb = R.VectorDatabase(cfg)
    vdb.create_index(D)
    vdb.index.add_device(rows)
    D0, I0 = (t.clone() for t in vdb.index.search_device(emb0, K))
    assert vdb.index.last_launch()["block_threads"] == 512          # the certified f16 scan
    for _ in range(40):
        e = fe.embed_clips(wave, offs)
        Dd, Ii = vdb.index.search_device(e, K)
        assert torch.equal(e, emb0) and torch.equal(Ii, I0) and torch.equal(Dd, D0)
