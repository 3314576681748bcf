"""CPU: the host-side mirror (no GPU): segmenter vs the reference golden vectors, mel filter bank, config, shard
bounds, and the C-ABI library: loads, exports every symbol include/radad_hip.h declares, argument errors surface as
the reference's exception types.  No compute entry point is called (there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
from radad_retrievalaugmenteddeepfakeaudiodetection_amd.feature_extractor import mel_filter_bank_slaney, synthetic_projection
from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    import torch
    c = R.Config()
    c.device = torch.device("cpu")
    return c


def _ramp(n):
    return (np.arange(n, dtype=np.float32) % 977) / np.float32(977.0) - np.float32(0.5)


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "radad_hip.h")).read()
    declared = set(re.findall(r"\b(radad_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"radad_embed_cfg", "radad_proj_weights"}
    assert declared == set(_lib.SIGNATURES), f"header vs binding mismatch: {declared ^ set(_lib.SIGNATURES)}"
    lib = _lib.load()                      # getattr on every declared name: raises if the .so lacks one
    assert lib.radad_abi_version() == 1
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name)


def test_library_is_the_in_tree_hip_build():
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, "radad_retrievalaugmenteddeepfakeaudiodetection_amd")
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_knn_f32" in blob and b"k_logmel" in blob    # device code objects are embedded


def test_abi_argument_errors_without_gpu():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.radad_knn_create(30, 0, 0, 0, C.byref(h)) == _lib.RADAD_EINVAL         # dim % 4
    assert b"multiple of 4" in lib.radad_last_error()
    assert lib.radad_knn_create(32, 7, 0, 0, C.byref(h)) == _lib.RADAD_EINVAL         # metric
    with pytest.raises(ValueError):
        _lib.check(lib.radad_knn_create(32, 7, 0, 0, C.byref(h)))
    assert lib.radad_knn_search(None, None, 1, 1, None, None, None) == _lib.RADAD_EINVAL
    assert lib.radad_topk_merge(0, None, None, 0, 1, 1, None, None, 0, None) == _lib.RADAD_EINVAL
    assert lib.radad_projection_workspace_bytes(4, 5, 512, 256, 128) > 0
    assert lib.radad_projection_workspace_bytes(-1, 5, 512, 256, 128) == -1


def test_segment_count_c_equals_python_rule(golden_dir):
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, "segmenter.npz"))
    for n in list(g["lengths"]) + [0, 1, 15999, 16000, 16001, 95999, 96000, 10 ** 7]:
        n = int(n)
        assert lib.radad_segment_count(n, 32000, 16000) == max(1, (n - 32000) // 16000 + 1)
    for n in g["lengths"]:
        assert lib.radad_segment_count(int(n), 32000, 16000) == int(g[f"n{int(n)}_count"])
    assert lib.radad_segment_count(10, 0, 5) == -1


def test_audio_segmenter_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "segmenter.npz"))
    seg = R.AudioSegmenter(_cfg())
    assert (seg.segment_length, seg.hop_length) == (int(g["segment_length"]), int(g["hop_length"]))
    for n in g["lengths"]:
        n = int(n)
        segs = seg.segment_audio(_ramp(n))
        assert len(segs) == int(g[f"n{n}_count"]) == seg.num_segments(n)
        assert str(segs[-1].dtype) == str(g[f"n{n}_dtype"])
        np.testing.assert_array_equal(np.stack([np.asarray(s[:8], np.float64) for s in segs]), g[f"n{n}_first8"])
        np.testing.assert_array_equal(np.stack([np.asarray(s[-8:], np.float64) for s in segs]), g[f"n{n}_last8"])
        np.testing.assert_array_equal(np.asarray([np.asarray(s, np.float64).sum() for s in segs]), g[f"n{n}_sum"])
    with pytest.raises(ValueError, match="Expected 1D audio array"):
        seg.segment_audio(np.zeros((2, 5), np.float32))
    # unpadded segments are views of the input, as in the reference
    a = _ramp(64000)
    assert np.shares_memory(seg.segment_audio(a)[1], a)


def test_segment_plan_is_the_csr_of_segment_audio():
    seg = R.AudioSegmenter(_cfg())
    lens = [100, 32000, 50000, 64000, 80001]
    offs, start, valid, clip_seg = seg.plan(lens)
    assert offs.tolist() == np.concatenate([[0], np.cumsum(lens)]).tolist()
    assert clip_seg.tolist() == np.concatenate([[0], np.cumsum([seg.num_segments(n) for n in lens])]).tolist()
    flat = np.concatenate([_ramp(n) for n in lens])
    for b, n in enumerate(lens):
        for j, s in enumerate(seg.segment_audio(_ramp(n))):
            i = clip_seg[b] + j
            np.testing.assert_array_equal(np.asarray(s[:valid[i]], np.float32), flat[start[i]:start[i] + valid[i]])


def test_mel_filter_bank_matches_hf(golden_dir):
    g = np.load(os.path.join(golden_dir, "frontend.npz"))
    np.testing.assert_allclose(mel_filter_bank_slaney(), g["mel_filters"], rtol=1e-12, atol=1e-15)


def test_config_and_selector():
    c = _cfg()
    assert (c.sample_rate, c.segment_length, c.segment_overlap, c.tpp_levels, c.tpp_pooling_type) == (16000, 2.0, 0.5, [1, 2, 4], "max")
    assert (c.vector_db_index_type, c.top_k, c.projection_hidden_dim, c.projection_output_dim) == ("L2", 5, 256, 128)
    c.update(top_k=7)
    assert c.top_k == 7
    with pytest.raises(ValueError, match="Invalid configuration parameter"):
        c.update(nope=1)
    c.feature_extractor_type = "bogus"
    with pytest.raises(ValueError):
        R.build_feature_extractor(c)
    c.feature_extractor_type = "wav2vec2"         # the reference's named extractors need their encoder on the LOCAL disk (never fetched)
    with pytest.raises(FileNotFoundError, match="never downloads"):
        R.build_feature_extractor(c)
    c.feature_extractor_type = "melproj"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        R.build_feature_extractor(c)              # device is cpu: the product path refuses, it does not fall back


def test_no_cpu_fallback_in_operators():
    import torch
    c = _cfg()
    c.feature_dim = 8
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        R.TemporalPyramidPooling(c).pool_features(torch.zeros(4, 8))
    p = R.ProjectionLayer(c, 16).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"), torch.no_grad():
        p(torch.zeros(1, 5, 16))
    # parameter names/shapes equal the reference's fused layout, so its state_dict loads unchanged
    sd = {k: tuple(v.shape) for k, v in p.state_dict().items()}
    assert sd == {"attention_score.weight": (256, 16), "attention_score.bias": (256,), "attention_final.weight": (1, 256),
                  "attention_final.bias": (1,), "cst_hidden.weight": (256, 16), "cst_hidden.bias": (256,),
                  "cst_output.weight": (16, 256), "cst_output.bias": (16,), "weight_sum.weight": (256, 16),
                  "weight_sum.bias": (256,), "normalization.weight": (256,), "normalization.bias": (256,),
                  "unified_embedding.weight": (128, 256), "unified_embedding.bias": (128,)}


def test_shard_bounds_partition():
    for n, w in ((10, 3), (1_000_000, 8), (7, 8), (0, 2)):
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_synthetic_projection_is_deterministic():
    w1, b1 = synthetic_projection(64, 5)
    w2, b2 = synthetic_projection(64, 5)
    np.testing.assert_array_equal(w1, w2)
    assert w1.shape == (80, 64) and b1.shape == (64,) and w1.dtype == np.float32


def test_oracle_is_not_imported_by_the_product():
    pkg = os.path.join(ROOT, "radad_retrievalaugmenteddeepfakeaudiodetection_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace("oracle/synth.py", ""), f


def test_load_follows_the_saved_index_type_not_a_stale_sidecar(tmp_path, monkeypatch):
    """IVF save -> flat save to the same directory -> load: the flat save removes the centroid sidecar of the earlier IVF
    store, and load() picks its branch from meta['index_type'] (it used to see the stale sidecar, build a flat index, raise
    'saved IVF store does not match config' into its own log-and-degrade handler and come back silently EMPTY).  Host logic
    only: the two index classes are stand-ins that write / read the native snapshot's magic (no GPU here)."""
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import vector_database as V

    class FakeFlat:
        def __init__(self, d, metric, device=0, id_base=0, store_f16=False):
            self.d, self.metric, self.id_base, self.store_f16, self.ntotal = d, metric, id_base, store_f16, 0
        def add(self, x):
            self.ntotal += len(x)
        def save(self, path):
            with open(path, "wb") as f:
                f.write(b"RADADKNN" + np.int64(self.ntotal).tobytes())
        def load(self, path, row0=0, n_rows=-1):
            raw = open(path, "rb").read()
            assert raw[:8] == b"RADADKNN"
            self.ntotal = int(np.frombuffer(raw[8:16], np.int64)[0]) if n_rows < 0 else n_rows
        @staticmethod
        def snapshot_info(path):
            raw = open(path, "rb").read()
            return {"d": 8, "metric": 0, "store_f16": False, "ntotal": int(np.frombuffer(raw[8:16], np.int64)[0])}

    class FakeIVF(FakeFlat):
        def __init__(self, d, nlist, device=0):
            FakeFlat.__init__(self, d, 0)
            self.nlist, self.is_trained = nlist, True
        def save(self, path):
            FakeFlat.save(self, path)
            with open(path + ".ivf.npz", "wb") as f:
                np.savez(f, kind="radad_ivf", ntotal=self.ntotal)
        def load(self, path):
            assert os.path.exists(path + ".ivf.npz")
            FakeFlat.load(self, path)

    monkeypatch.setattr(V, "HipFlatIndex", FakeFlat)
    monkeypatch.setattr(V, "HipIVFFlatIndex", FakeIVF)
    c = _cfg()
    c.vector_db_path = str(tmp_path)
    rows = np.zeros((12, 8), np.float32)
    names = [f"f{i}.wav" for i in range(12)]

    c.vector_db_index_type = "IVF"
    a = V.VectorDatabase(c)
    a.add_vectors(rows, names, [0] * 12, {})
    a.save()
    assert os.path.exists(a.db_path + ".ivf.npz")
    b = V.VectorDatabase(c)
    b.load()
    assert isinstance(b.index, FakeIVF) and b.index.ntotal == 12

    c.vector_db_index_type = "L2"                      # rebuild as a flat store, save to the same path
    f = V.VectorDatabase(c)
    f.add_vectors(rows[:7], names[:7], [1] * 7, {})
    f.save()
    assert not os.path.exists(f.db_path + ".ivf.npz"), "the flat save must drop the earlier IVF store's sidecar"
    g = V.VectorDatabase(c)
    g.load()
    assert type(g.index) is FakeFlat and g.index.ntotal == 7 and len(g.vector_paths) == 7
    s = V.VectorDatabase(c)
    s.load(shard=(1, 2))                               # a row shard of the flat snapshot loads as well
    assert type(s.index) is FakeFlat and s.index.ntotal in (3, 4) and len(s.vector_paths) == s.index.ntotal

    # even with a stale sidecar lying around (written by a build without the fix), the recorded type decides
    with open(f.db_path + ".ivf.npz", "wb") as fh:
        np.savez(fh, kind="radad_ivf", ntotal=12)
    g2 = V.VectorDatabase(c)
    g2.load()
    assert type(g2.index) is FakeFlat and g2.index.ntotal == 7


def test_clip_chunking_covers_every_frame_once_and_fits_the_planes():
    """the work list of k_logmel_h_clip (csrc/logmel_h.inc: clip_chunking, shared by the kernel, the plan kernel and the host; host
    arithmetic, no GPU): for every (segments, frames per segment, segment hop in frames) the chunks hold each interior frame of the
    clip and each of its 3 S edge frames exactly once, no chunk exceeds 128 frame slots (four 32-frame waves) or the LDS planes
    (20 272 halfs: the interior frames' samples at a stride of 168 halfs per frame + 424 halfs per edge window)."""
    lib = _lib.load()
    out = (C.c_int32 * 5)()
    for T, H in ((200, 100), (200, 50), (100, 50), (50, 40), (200, 150), (8, 4), (223, 1), (64, 63)):
        for S in list(range(1, 40)) + [97, 500]:
            _lib.check(lib.radad_embed_clip_chunks(S, T, H, out))
            n_full, r, e_tail, n_edge_chunks, total = (int(v) for v in out)
            ni, E = (S - 1) * H + T - 3, 3 * S
            assert n_full * 104 + r == ni and 0 <= r < 104                              # interior frames: full chunks + the tail's
            assert total == n_full + 1 + n_edge_chunks
            assert 0 <= e_tail <= E and r + e_tail <= 128
            planes_r = 0 if r == 0 else -(-((160 * (r - 1) + 400) + 8 * ((160 * (r - 1) + 400) // 160) + 8) // 16) * 16
            assert planes_r + 424 * e_tail <= 20272, (T, H, S)
            rest = E - e_tail                                                            # edge frames left for the edge-only chunks
            assert n_edge_chunks == -(-rest // 32) and (rest == 0) == (n_edge_chunks == 0)
    _lib.check(lib.radad_embed_clip_chunks(0, 200, 100, out))
    assert list(out) == [0, 0, 0, 0, 0]
    assert lib.radad_embed_clip_chunks(3, 4, 2, out) == _lib.RADAD_EINVAL               # fewer than 8 frames per segment: not this path


def test_scan_geometry_is_chosen_by_cost():
    """radad_knn_scan_geometry (host arithmetic of the tile scan's launch, no device): chunks are whole tiles and cover the store; the
    launch never costs more (rounds of 256 workgroups x tiles per chunk) than round 4's fixed two rounds' worth of chunks; and the cases
    the choice was made for: BASELINE config 2 runs ONE round, configs 4 / 5 at full size exactly five."""
    lib = _lib.load()

    def geo(n, nq):
        qt, ch, rows = C.c_int(), C.c_int(), C.c_int64()
        _lib.check(lib.radad_knn_scan_geometry(n, nq, C.byref(qt), C.byref(ch), C.byref(rows)))
        return qt.value, ch.value, rows.value

    def cost(n, qt, rows):
        tiles, per = -(-n // 256), rows // 256
        return -(-(qt * -(-tiles // per)) // 256) * per

    def old(n, nq):                                   # round 4: ceil(512 / query tiles) chunks, rounded up to a multiple of 8
        qt, tiles = -(-nq // 256), -(-n // 256)
        want = min(max(8, -(-min(-(-512 // qt), tiles) // 8) * 8), 1024)
        return qt, want, -(-tiles // want) * 256

    for n, nq in ((100_000, 1024), (1_000_000, 1024), (25_423, 256), (1_300_000, 700), (10_000_000, 10_240), (50_000_000, 10_240),
                  (20_000, 300), (125_000, 8192), (200_000, 512), (1_000_000, 256), (1_250_000, 10_240), (16_384, 17), (999_937, 1023)):
        qt, ch, rows = geo(n, nq)
        assert qt == -(-nq // 256) and ch % 8 == 0 and ch >= 8 and rows % 256 == 0 and ch * rows >= n, (n, nq, qt, ch, rows)
        oq, och, orows = old(n, nq)
        assert cost(n, qt, rows) <= cost(n, oq, orows), (n, nq, (qt, ch, rows), (oq, och, orows))
    assert geo(100_000, 1024) == (4, 64, 7 * 256)                      # config 2: 224 workgroups, one round of 7 tiles (was 2 x 4)
    assert geo(1_000_000, 1024) == (4, 128, 31 * 256)                  # the headline: two rounds of 31 (one of 62 costs the same)
    assert geo(10_000_000, 10_240) == (40, 32, 1221 * 256)             # config 4: 1280 workgroups = 5 rounds (was 640 = 2.5 -> 3)
    assert geo(50_000_000, 10_240)[1] == 32
    assert lib.radad_knn_scan_geometry(0, 1, None, None, None) == _lib.RADAD_EINVAL
