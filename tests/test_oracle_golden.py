"""CPU: pin the oracle (oracle/radad_oracle.py) to golden vectors produced by the reference's own modules
and by the HuggingFace front-ends the reference calls (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import radad_oracle as O
from oracle import synth


@pytest.fixture(scope="module")
def g_seg(golden_dir):
    return np.load(os.path.join(golden_dir, "segmenter.npz"))


def _ramp(n):
    return (np.arange(n, dtype=np.float32) % 977) / np.float32(977.0) - np.float32(0.5)


def test_segment_lengths_match_reference(g_seg):
    seg, hop = O.segment_lengths()
    assert (seg, hop) == (int(g_seg["segment_length"]), int(g_seg["hop_length"])) == (32000, 16000)


def test_segmenter_matches_reference(g_seg):
    seg, hop = O.segment_lengths()
    for n in g_seg["lengths"]:
        n = int(n)
        segs = O.segment_audio(_ramp(n), seg, hop)
        assert len(segs) == int(g_seg[f"n{n}_count"]) == O.segment_count(n, seg, hop)
        assert str(segs[-1].dtype) == str(g_seg[f"n{n}_dtype"])          # float64 promotion when padded (segmenter.py:36)
        assert all(len(s) == seg for s in segs)
        np.testing.assert_array_equal(np.stack([np.asarray(s[:8], np.float64) for s in segs]), g_seg[f"n{n}_first8"])
        np.testing.assert_array_equal(np.stack([np.asarray(s[-8:], np.float64) for s in segs]), g_seg[f"n{n}_last8"])
        np.testing.assert_array_equal(np.asarray([np.asarray(s, np.float64).sum() for s in segs]), g_seg[f"n{n}_sum"])


def test_segmenter_known_counts():
    seg, hop = O.segment_lengths()
    # tail beyond the last full window is dropped (segmenter.py:25): 70 001 -> 3, 80 000 -> 4
    assert [O.segment_count(n, seg, hop) for n in (0, 100, 32000, 47999, 48000, 64000, 70001, 80000)] == [1, 1, 1, 1, 2, 3, 3, 4]
    with pytest.raises(ValueError):
        O.segment_audio(np.zeros((2, 10)), seg, hop)


def test_segment_plan_equals_list_form():
    seg, hop = O.segment_lengths()
    lens = [100, 32000, 50000, 64000, 80001]
    clip, start, valid, offs = O.segment_plan(lens, seg, hop)
    assert offs.tolist() == np.cumsum([0] + [O.segment_count(n, seg, hop) for n in lens]).tolist()
    for b, n in enumerate(lens):
        segs = O.segment_audio(_ramp(n), seg, hop)
        for j, s in enumerate(segs):
            i = offs[b] + j
            np.testing.assert_array_equal(np.asarray(s[:valid[i]], np.float32), _ramp(n)[start[i]:start[i] + valid[i]])
            assert np.all(np.asarray(s[valid[i]:]) == 0)


def test_tpp_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "pooling.npz"))
    for key in g["cases"]:
        key = str(key)
        mode, levels, shape = key.split("_")
        levels = [int(v) for v in levels.split("-")]
        y = O.tpp(g[key + "_x"], levels, mode)
        assert y.shape[0] == O.tpp_output_dim(levels, g[key + "_x"].shape[1])
        if mode == "max":
            np.testing.assert_array_equal(y.astype(np.float32), g[key + "_y"])   # max is order independent: bit exact
        else:
            np.testing.assert_allclose(y, g[key + "_y"], rtol=0, atol=2e-6)
    for S in (1, 2, 3):
        np.testing.assert_allclose(O.segment_mean(list(g[f"segmean{S}_x"])), g[f"segmean{S}_y"], rtol=0, atol=1e-6)


def test_adaptive_bins_overlap_rule():
    assert O.adaptive_bins(99, 2) == [(0, 50), (49, 99)]
    assert O.adaptive_bins(99, 4) == [(0, 25), (24, 50), (49, 75), (74, 99)]
    assert O.adaptive_bins(3, 4) == [(0, 1), (0, 2), (1, 3), (2, 3)]


def _proj_shapes(D, H=256, Oo=128):
    return {"attention_score.weight": (H, D), "attention_score.bias": (H,), "attention_final.weight": (1, H),
            "attention_final.bias": (1,), "cst_hidden.weight": (H, D), "cst_hidden.bias": (H,),
            "cst_output.weight": (D, H), "cst_output.bias": (D,), "weight_sum.weight": (H, D), "weight_sum.bias": (H,),
            "normalization.weight": (H,), "normalization.bias": (H,), "unified_embedding.weight": (Oo, H),
            "unified_embedding.bias": (Oo,)}


def test_projection_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "projection.npz"))
    for D in (512, 3584):
        shapes = _proj_shapes(D)
        assert list(shapes) == [str(n) for n in g[f"proj{D}_names"]]
        sd = synth.fill_state_dict(shapes, int(g[f"proj{D}_seed"]))
        y = O.projection_forward(g[f"proj{D}_x"], sd)
        np.testing.assert_allclose(y, g[f"proj{D}_y"], rtol=0, atol=2e-5)


def test_radad_shell_projection_and_fuse(golden_dir):
    """radad_model.py:38-39: proj = ProjectionLayer(neigh); fused = fuse(cat[tpp, proj])."""
    g = np.load(os.path.join(golden_dir, "projection.npz"))
    names = [str(n) for n in g["radad_names"]]
    shapes = {n: tuple(int(v) for v in str(s).split(",") if v) for n, s in zip(names, g["radad_shapes"])}
    sd = synth.fill_state_dict(shapes, int(g["radad_seed"]))
    psd = {k[len("projection_layer."):]: v for k, v in sd.items() if k.startswith("projection_layer.")}
    proj = O.projection_forward(g["radad_x"], psd)
    np.testing.assert_allclose(proj, g["radad_proj"], rtol=0, atol=2e-5)
    fused = np.concatenate([g["radad_t"].astype(np.float64), proj], axis=1) @ sd["fuse.weight"].astype(np.float64).T + sd["fuse.bias"]
    np.testing.assert_allclose(fused, g["radad_fused"], rtol=0, atol=5e-5)


@pytest.fixture(scope="module")
def g_fe(golden_dir):
    return np.load(os.path.join(golden_dir, "frontend.npz"))


def test_zero_mean_unit_var_matches_hf(g_fe):
    y = O.zero_mean_unit_var(g_fe["segments"])
    np.testing.assert_allclose(y, g_fe["w2v_normalized"], rtol=0, atol=5e-6)
    assert abs(y.mean()) < 1e-9 and abs(y.var(axis=1) - 1).max() < 1e-5


def test_mel_filter_bank_matches_hf(g_fe):
    fb = O.mel_filter_bank()
    assert fb.shape == (201, 80)
    np.testing.assert_allclose(fb, g_fe["mel_filters"], rtol=1e-12, atol=1e-15)


def test_log_mel_self_mode_matches_hf(g_fe):
    for s in range(2):
        lm = O.log_mel(g_fe["segments"][s])               # [200, 80]
        assert lm.shape == (200, 80)
        np.testing.assert_allclose(lm.T, g_fe["whisper_self"][s], rtol=0, atol=2e-4)
        assert np.abs(lm.T - g_fe["whisper_self"][s]).mean() < 5e-6


def test_log_mel_padded_mode_matches_hf(g_fe):
    assert bool(g_fe["whisper_padded_const_ok"])
    for s in range(2):
        lm = O.log_mel(g_fe["segments"][s], padded_samples=480000)    # [3000, 80]
        assert lm.shape == (3000, 80)
        np.testing.assert_allclose(lm[:208].T, g_fe["whisper_padded_first208"][s], rtol=0, atol=2e-4)
        # frames whose window lies entirely in the zero padding are one constant row
        np.testing.assert_allclose(lm[208:], np.broadcast_to(g_fe["whisper_padded_tail"][s], (3000 - 208, 80)), rtol=0, atol=1e-6)
        assert np.all(lm[202:] == lm[-1])


def test_knn_oracle_semantics():
    rng = np.random.default_rng(0)
    db = rng.standard_normal((300, 16)).astype(np.float32)
    q = rng.standard_normal((5, 16)).astype(np.float32)
    d, i = O.knn(db, q, 7, "L2")
    full = ((q[:, None, :].astype(np.float64) - db[None].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_array_equal(i, np.argsort(full, axis=1, kind="stable")[:, :7])
    np.testing.assert_allclose(d, np.sort(full, axis=1)[:, :7], rtol=1e-12)
    d, i = O.knn(db, q, 7, "IP")
    ip = q.astype(np.float64) @ db.astype(np.float64).T
    np.testing.assert_array_equal(i, np.argsort(-ip, axis=1, kind="stable")[:, :7])
    # ties resolve to the lower index; k is clamped to ntotal; k<=0 is empty (vector_database.py:169-172)
    dup = np.concatenate([db[:3], db[:3]])
    _, i = O.knn(dup, db[:1], 6, "L2")
    assert i[0, 0] == 0 and i[0, 1] == 3
    assert O.knn(db[:4], q, 9, "L2")[1].shape == (5, 4)
    assert O.knn(db, q, 0, "L2")[0].shape == (5, 0)
    # cosine = IP on normalised rows (vector_database.py:97,100-105)
    dc, ic = O.knn(db, q, 5, "COSINE")
    nd, nq = O.maybe_normalize(db, True), O.maybe_normalize(q, True)
    np.testing.assert_array_equal(ic, O.knn(nd, nq, 5, "IP")[1])
    assert np.all(np.abs(dc) <= 1 + 1e-12)


def test_c_oracle_equals_numpy_oracle(knn_oracle_lib):
    from conftest import c_knn
    rng = np.random.default_rng(1)
    db = rng.standard_normal((5000, 64)).astype(np.float32)
    db[4000:4010] = db[10:20]        # exact duplicates: tie -> lower id
    q = np.concatenate([rng.standard_normal((30, 64)).astype(np.float32), db[10:13]])
    for metric in ("L2", "IP"):
        d, i = c_knn(knn_oracle_lib, db, q, 12, metric, id_base=1000)
        od, oi = O.knn(db, q, 12, metric)
        np.testing.assert_array_equal(i, oi + 1000)
        np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-9)
    d, i = c_knn(knn_oracle_lib, db[:5], q, 8, "L2")
    assert np.all(i[:, 5:] == -1) and np.all(np.isinf(d[:, 5:]))


def test_merge_topk():
    rng = np.random.default_rng(2)
    db = rng.standard_normal((900, 8)).astype(np.float32)
    q = rng.standard_normal((11, 8)).astype(np.float32)
    for metric in ("L2", "IP"):
        parts = [O.knn(db[s:s + 300], q, 6, metric) for s in (0, 300, 600)]
        d = [p[0] for p in parts]
        i = [p[1] + s for p, s in zip(parts, (0, 300, 600))]
        md, mi = O.merge_topk(d, i, 6, metric)
        od, oi = O.knn(db, q, 6, metric)
        np.testing.assert_array_equal(mi, oi)
        np.testing.assert_allclose(md, od)


def test_retrieve_postprocess_matches_pipeline_rules():
    D, K = 4, 3
    stored = np.arange(40, dtype=np.float32).reshape(10, D)
    paths = [f"/data/a/f{i}.wav" for i in range(10)]
    labels = [i % 2 for i in range(10)]
    idxs = np.asarray([[0, 1, 2, 3, 4], [5, 6, 7, 8, 9]])
    dists = np.asarray([[0.0, .1, .2, .3, .4], [0.0, .1, .2, .3, .4]], np.float32)
    v, l, p, d = O.retrieve_postprocess(dists, idxs, stored, paths, labels, K, D, query_paths=["/x/f0.wav", "/y/f6.wav"])
    # basenames of the WHOLE query batch are excluded from every row (pipeline.py:463,497-499)
    assert [os.path.basename(x) for x in p[0]] == ["f1.wav", "f2.wav", "f3.wav"]
    assert [os.path.basename(x) for x in p[1]] == ["f5.wav", "f7.wav", "f8.wav"]
    np.testing.assert_array_equal(v[1, 1], stored[7])
    # fewer than K survivors -> zero / 0.0 / "" / NaN padding (pipeline.py:511-515)
    v, l, p, d = O.retrieve_postprocess(dists[:, :2], idxs[:, :2], stored, paths, labels, K, D, query_paths=["f0.wav"])
    assert p[0] == [paths[1], "", ""] and np.all(v[0, 1:] == 0) and np.isnan(d[0, 1:]).all() and l[0, 1] == 0.0
