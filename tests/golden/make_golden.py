"""Generate the golden vectors under tests/golden/ by IMPORTING the reference's own modules
(/root/reference: segmenter.py, pooling.py, projection.py, radad_model.py, and -- round 5 -- pipeline.py's two hot-path
methods and vector_database.py's shell around the faiss index) and the HuggingFace front-ends the
reference's extractors call (default constructors, no from_pretrained / no network).

Run here (the container that has /root/reference); the GPU box never sees the reference:
    python tests/golden/make_golden.py
Only DATA is written (inputs by seed or by value, outputs by value) -- no reference source.

The reference's config.py imports torchaudio / faiss / librosa at module top (config.py:3,14,16) although the
modules imported here never use them; those three names are registered as empty modules so the import
resolves (this is the procedure SURVEY.md section 8c records).  faiss itself is NOT emulated: the kNN has no
golden vectors from the reference (its arithmetic lives in faiss, absent here) -- see oracle/__init__.py.
pipeline.npz (round 5): DeepfakeDetectionPipeline.process_audio_batch / .retrieve_similar_vectors and
VectorDatabase._maybe_normalize / .add_vectors / .search_batch are run on instances made with __new__ (no __init__: that would
construct the pretrained extractor / faiss resources) whose collaborators are the reference's own AudioSegmenter and
TemporalPyramidPooling, a seeded stand-in extractor, and a RECORDING index (it stores what the reference hands to
index.add / index.search and returns seeded lists: faiss arithmetic is not imitated).
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
    # the unfused parameter layout (fuse_attention_ops=False, projection.py:32-47): same arithmetic, nn.Sequential parameter names
    cfg.fuse_attention_ops = False
    D, B, seed = 512, 3, 5400
    layer = ref_projection.ProjectionLayer(cfg, D).eval()
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in layer.state_dict().items()}, seed)
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = torch.from_numpy(synth.rows(0, B * 5, D, seed + 99).reshape(B, 5, D))
    with torch.no_grad():
        y = layer(x)
    g["projU_seed"], g["projU_names"] = np.asarray(seed), np.asarray(list(sd.keys()))
    g["projU_shapes"] = np.asarray([",".join(map(str, v.shape)) for v in sd.values()])
    g["projU_x"], g["projU_y"] = x.numpy(), y.numpy()
    cfg.fuse_attention_ops = True
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
    pipeline_fixture(ref_config, ref_segmenter, ref_pooling)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


# the stand-in encoder of the a4 fixture: frames of 320 samples -> tanh(frame @ W), float32 (any deterministic [T, F] map would do;
# the oracle and the GPU test apply the same function through the extractor protocol)
A4_F, A4_HOP, A4_T = 8, 320, 99


def a4_extract(segment, w):
    x = np.asarray(segment, np.float32)[:A4_T * A4_HOP].reshape(A4_T, A4_HOP)
    return np.tanh(x @ w).astype(np.float32)


def pipeline_fixture(ref_config, ref_segmenter, ref_pooling):
    """a4 (pipeline.py:392-414), a7 (pipeline.py:449-532), K7 + the index shell (vector_database.py:100-188), by RUNNING them."""
    import torch
    import pipeline as ref_pipeline
    import vector_database as ref_vdb
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import synth
    g = {}

    # ---- a4: process_audio_batch ---------------------------------------------------------------------------------
    cfg = ref_config.Config()
    cfg.device = torch.device("cpu")
    cfg.tpp_levels, cfg.tpp_pooling_type, cfg.feature_dim = [1, 2, 4], "max", A4_F
    lengths = [20000, 32000, 48000, 64000, 70001, 31999]          # < one segment (float64 padding), 1, 2, 3, 3 (tail dropped), padded
    w = synth.rows(0, A4_HOP, A4_F, 8100) * np.float32(0.05)
    waves = {f"/data/c{i}.wav": synth.audio(i, 1, n, 8101)[0] for i, n in enumerate(lengths)}

    class FakeExtractor:
        feature_dim = A4_F

        def extract_features(self, segments):
            return [torch.from_numpy(a4_extract(s, w)) for s in segments]

    class FakeDataset:
        def load_audio(self, path):
            return waves[path]

    for mode in ("max", "avg"):
        cfg.tpp_pooling_type = mode
        pipe = ref_pipeline.DeepfakeDetectionPipeline.__new__(ref_pipeline.DeepfakeDetectionPipeline)
        pipe.config, pipe.device = cfg, torch.device("cpu")
        pipe.audio_segmenter = ref_segmenter.AudioSegmenter(cfg)
        pipe.feature_extractor = FakeExtractor()
        pipe.tpp = ref_pooling.TemporalPyramidPooling(cfg)
        out = pipe.process_audio_batch(list(waves.keys()), FakeDataset())
        g[f"a4_{mode}_out"] = out.numpy()
    g["a4_lengths"], g["a4_w_seed"], g["a4_audio_seed"] = np.asarray(lengths), np.asarray(8100), np.asarray(8101)
    g["a4_levels"] = np.asarray([1, 2, 4])
    failed = False
    try:                                                               # pipeline.py:398-399
        class NoneDataset:
            def load_audio(self, path):
                return None
        pipe.process_audio_batch(["/x.wav"], NoneDataset())
    except RuntimeError as e:
        failed = "Failed to load" in str(e)
    g["a4_none_raises_runtime_error"] = np.asarray(failed)

    # ---- a7: retrieve_similar_vectors ------------------------------------------------------------------------------
    K, D, N, B = 5, 56, 40, 6
    cfg.top_k = K
    cfg.tpp_levels, cfg.feature_dim = [1, 2, 4], 8                 # tpp.get_output_dim() = 56
    stored = synth.rows(0, N, D, 8200)
    # duplicate basenames across directories; the last path is the one a -1 id wraps to (pipeline.py:495)
    paths = [f"/train/{'a' if i % 2 else 'b'}/f{i % 30}.wav" for i in range(N)]
    labels = [float(i % 2) for i in range(N)]
    rng = np.random.default_rng(8201)
    query_paths = ["/eval/f3.wav", "/eval/q1.wav", "/other/f7.wav", "/eval/q3.wav", "/eval/f12.wav", "/eval/q5.wav"]

    def make_hits(kk):
        idxs = np.stack([rng.permutation(N)[:kk] for _ in range(B)]).astype(np.int64)
        dists = np.sort(rng.random((B, kk)).astype(np.float32), axis=1)
        return dists, idxs
    d15, i15 = make_hits(K + 10)
    i15[0, :4] = [3, 33, 7, 37]                                    # row 0: leading hits are the batch's own basenames (f3, f7 twice each)
    i15[1, :] = [3, 33, 7, 37, 12, 3, 33, 7, 37, 12, 3, 33, 7, 9, 12]   # row 1: only ONE survivor -> padded to K
    i15[2, 6:] = -1                                                # row 2: faiss's unfilled slots
    i15[3, :] = -1                                                 # row 3: nothing found at all
    d5, i5 = make_hits(K)
    i5[4, 2] = -1
    train_ids = {"f1.wav", "f2.wav", "f29.wav", "f39.wav"}

    class RecordingSearch:
        def __init__(self, dists, idxs, fail=False):
            self.dists, self.idxs, self.fail, self.calls = dists, idxs, fail, []

        def __call__(self, q, k=None):
            self.calls.append((np.asarray(q).shape, np.asarray(q).dtype.str, int(k)))
            if self.fail:
                raise RuntimeError("search failed")
            return self.dists[:, :k], self.idxs[:, :k]

    def run(search, ntotal, **kw):
        pipe = ref_pipeline.DeepfakeDetectionPipeline.__new__(ref_pipeline.DeepfakeDetectionPipeline)
        pipe.config, pipe.device = cfg, torch.device("cpu")
        pipe.tpp = ref_pooling.TemporalPyramidPooling(cfg)
        pipe.training_file_ids = set(train_ids)
        pipe.vector_db = types.SimpleNamespace(
            index=types.SimpleNamespace(ntotal=ntotal, reconstruct=lambda i: stored[i].copy()),
            vector_paths=paths, vector_labels=labels, search_batch=search)
        q = torch.from_numpy(synth.rows(0, B, D, 8202))
        return pipe.retrieve_similar_vectors(q, **kw)

    cases = {
        "self_paths": dict(search=(d15, i15), kw=dict(query_paths=query_paths, exclude_self=True)),
        "self_train_ids": dict(search=(d15, i15), kw=dict(query_paths=None, exclude_self=True)),
        "no_exclusion": dict(search=(d5, i5), kw=dict(query_paths=query_paths, exclude_self=False)),
        "search_raises": dict(search=(d15, i15), fail=True, kw=dict(query_paths=query_paths, exclude_self=True)),
        "empty_index": dict(search=(d15, i15), ntotal=0, kw=dict(query_paths=query_paths, exclude_self=True)),
    }
    for name, c in cases.items():
        s = RecordingSearch(*c["search"], fail=c.get("fail", False))
        vec, lbl, pth, dist = run(s, c.get("ntotal", N), return_info=True, return_distances=True, **c["kw"])
        g[f"a7_{name}_vec"], g[f"a7_{name}_lbl"], g[f"a7_{name}_dist"] = vec.numpy(), lbl.numpy(), dist.numpy()
        g[f"a7_{name}_paths"] = np.asarray(pth)
        g[f"a7_{name}_search_k"] = np.asarray([c_[2] for c_ in s.calls], np.int64)
        g[f"a7_{name}_query_dtype"] = np.asarray([c_[1] for c_ in s.calls])
        # the other three arities return the same tensors
        s2 = RecordingSearch(*c["search"], fail=c.get("fail", False))
        r2 = run(s2, c.get("ntotal", N), **c["kw"])
        r3 = run(s2, c.get("ntotal", N), return_info=True, **c["kw"])
        r4 = run(s2, c.get("ntotal", N), return_distances=True, **c["kw"])
        ar = [len(r2), len(r3), len(r4)]
        same = (torch.equal(r2[0], vec) and torch.equal(r2[1], lbl) and torch.equal(r3[0], vec) and r3[2] == pth and
                torch.equal(r4[0], vec) and torch.equal(torch.nan_to_num(r4[2], nan=-7.0), torch.nan_to_num(dist, nan=-7.0)))
        g[f"a7_{name}_arities"] = np.asarray(ar + [int(same)])
    g["a7_K"], g["a7_D"], g["a7_N"], g["a7_B"] = map(np.asarray, (K, D, N, B))
    g["a7_stored_seed"], g["a7_query_seed"] = np.asarray(8200), np.asarray(8202)
    g["a7_paths"], g["a7_labels"] = np.asarray(paths), np.asarray(labels, np.float32)
    g["a7_query_paths"], g["a7_training_file_ids"] = np.asarray(query_paths), np.asarray(sorted(train_ids))
    g["a7_d15"], g["a7_i15"], g["a7_d5"], g["a7_i5"] = d15, i15, d5, i5

    # ---- K7 + the shell around the index: vector_database.py:100-105, 108-157, 159-188 ---------------------------------------
    class RecordingIndex:
        """NOT faiss: stores what the reference hands over.  search returns the first k ids (the fixture pins the ARGUMENTS)."""
        def __init__(self):
            self.batches, self.searches = [], []
            self.is_trained = True

        @property
        def ntotal(self):
            return int(sum(len(b) for b in self.batches))

        def add(self, x):
            assert x.dtype == np.float32 and x.flags["C_CONTIGUOUS"]
            self.batches.append(np.array(x, copy=True))

        def search(self, q, k):
            assert q.dtype == np.float32 and q.flags["C_CONTIGUOUS"]
            self.searches.append((np.array(q, copy=True), int(k)))
            return (np.zeros((len(q), k), np.float32), np.tile(np.arange(k, dtype=np.int64), (len(q), 1)))

    rows = synth.rows(0, 23, 24, 8300) * np.float32(3.0)
    rows[5] = 0.0                                                   # a zero row: 0 / (0 + 1e-12) = 0
    rows[6] *= np.float32(1e-20)                                    # a row whose norm underflows towards the 1e-12 guard
    rows[7] *= np.float32(1e4)
    g["k7_rows"] = rows
    for cosine in (False, True):
        vdb = ref_vdb.VectorDatabase.__new__(ref_vdb.VectorDatabase)
        vdb._cosine = cosine
        vdb.gpu_resources = vdb.gpu_index = None                 # (what __del__ looks at)
        g[f"k7_norm_{int(cosine)}"] = np.asarray(vdb._maybe_normalize(rows.copy()))
        g[f"k7_norm_{int(cosine)}_dtype"] = np.asarray(str(np.asarray(vdb._maybe_normalize(rows.copy())).dtype))
        # add_vectors in batches of 10 (vector_add_batch_size), then search_batch with k above / at / below ntotal, 1-D query
        vdb.config = types.SimpleNamespace(vector_add_batch_size=10, top_k=5)
        vdb.index = RecordingIndex()
        vdb.vector_paths, vdb.vector_labels, vdb.vector_metadata = [], [], {}
        p_ = [f"/t/r{i}.wav" for i in range(23)]
        vdb.add_vectors(rows.copy(), p_, list(range(23)), {"speaker_id": [f"s{i % 3}" for i in range(23)]})
        g[f"shell_{int(cosine)}_added"] = np.concatenate(vdb.index.batches)
        g[f"shell_{int(cosine)}_batch_sizes"] = np.asarray([len(b) for b in vdb.index.batches])
        g[f"shell_{int(cosine)}_paths_ok"] = np.asarray(vdb.vector_paths == p_ and vdb.vector_labels == list(range(23)) and
                                                        vdb.vector_metadata["speaker_id"] == [f"s{i % 3}" for i in range(23)])
        q = synth.rows(0, 4, 24, 8301)
        ks = []
        for k_req, qq in ((None, q), (15, q), (23, q), (40, q), (3, q[0])):
            d_, i_ = vdb.search_batch(qq.copy(), k=k_req)
            ks.append([-1 if k_req is None else k_req, vdb.index.searches[-1][1], d_.shape[0], d_.shape[1]])
        g[f"shell_{int(cosine)}_k"] = np.asarray(ks)                 # [requested, passed to index.search, rows, cols]
        g[f"shell_{int(cosine)}_q_passed"] = vdb.index.searches[1][0]
        g[f"shell_{int(cosine)}_q1d_passed"] = vdb.index.searches[-1][0]
        # an index that holds nothing: ([B, 0] float32, [B, 0] int64) (vector_database.py:169-172)
        vdb.index = RecordingIndex()
        d_, i_ = vdb.search_batch(q.copy(), k=5)
        g[f"shell_{int(cosine)}_empty"] = np.asarray([d_.shape[0], d_.shape[1], i_.shape[0], i_.shape[1], int(d_.dtype == np.float32),
                                                      int(i_.dtype == np.int64), len(vdb.index.searches)])
        vdb.index = None
        try:
            vdb.search_batch(q, k=5)
            g[f"shell_{int(cosine)}_none_raises"] = np.asarray(False)
        except ValueError:
            g[f"shell_{int(cosine)}_none_raises"] = np.asarray(True)
    g["k7_query"] = synth.rows(0, 4, 24, 8301)
    np.savez_compressed(os.path.join(OUT, "pipeline.npz"), **g)


if __name__ == "__main__":
    main()
