"""GPU parity: ProjectionLayer forward vs the reference golden vectors, and the two hot pipeline methods
(process_audio_batch / retrieve_similar_vectors) vs the oracle restatement of pipeline.py:392-414,449-532."""
import os

import numpy as np
import pytest

from oracle import radad_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu


def _proj_shapes(D, H=256, Oo=128):
    return {"attention_score.weight": (H, D), "attention_score.bias": (H,), "attention_final.weight": (1, H),
            "attention_final.bias": (1,), "cst_hidden.weight": (H, D), "cst_hidden.bias": (H,),
            "cst_output.weight": (D, H), "cst_output.bias": (D,), "weight_sum.weight": (H, D), "weight_sum.bias": (H,),
            "normalization.weight": (H,), "normalization.bias": (H,), "unified_embedding.weight": (Oo, H),
            "unified_embedding.bias": (Oo,)}


@pytest.mark.parametrize("D", [512, 3584])
def test_projection_matches_reference_golden(gpu, golden_dir, D):
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    g = np.load(os.path.join(golden_dir, "projection.npz"))
    cfg = R.Config()
    cfg.update(device=gpu)
    layer = R.ProjectionLayer(cfg, D).eval()
    sd = synth.fill_state_dict(_proj_shapes(D), int(g[f"proj{D}_seed"]))
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})         # the reference's own key names
    with torch.no_grad():
        y = layer(torch.from_numpy(g[f"proj{D}_x"]).to(gpu)).cpu().numpy()
    np.testing.assert_allclose(y, g[f"proj{D}_y"], rtol=0, atol=1e-4)               # vs the reference module's output
    np.testing.assert_allclose(y, O.projection_forward(g[f"proj{D}_x"], sd), rtol=0, atol=1e-4)
    # a batch that is not a multiple of any tile, and forward_batch (projection.py:119-122)
    x = torch.from_numpy(synth.rows(0, 37 * 5, D, 77).reshape(37, 5, D)).to(gpu)
    with torch.no_grad():
        y2 = layer(x).cpu().numpy()
        y3 = layer.forward_batch([x[i] for i in range(37)]).cpu().numpy()
    np.testing.assert_allclose(y2, O.projection_forward(x.cpu().numpy(), sd), rtol=0, atol=1e-4)
    np.testing.assert_array_equal(y2, y3)
    layer.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        layer(x)


@pytest.mark.parametrize("fuse_here", [True, False])
def test_projection_loads_the_unfused_parameter_layout(gpu, golden_dir, fuse_here):
    """projection.py:32-47: with fuse_attention_ops=False the reference keeps the same four Linear layers inside two nn.Sequential
    (attention_score.{0,2}, cst_attention.{0,2}).  Such a state_dict must load -- into a layer of either configuration -- and give
    the reference module's output; a layer configured unfused writes those names back."""
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    g = np.load(os.path.join(golden_dir, "projection.npz"))
    names = [str(n) for n in g["projU_names"]]
    shapes = {n: tuple(int(v) for v in str(sh).split(",")) for n, sh in zip(names, g["projU_shapes"])}
    assert "attention_score.2.weight" in names and "cst_attention.0.bias" in names
    sd = synth.fill_state_dict(shapes, int(g["projU_seed"]))
    cfg = R.Config()
    cfg.update(device=gpu, fuse_attention_ops=fuse_here)
    layer = R.ProjectionLayer(cfg, 512).eval()
    res = layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert not res.missing_keys and not res.unexpected_keys
    with torch.no_grad():
        y = layer(torch.from_numpy(g["projU_x"]).to(gpu)).cpu().numpy()
    np.testing.assert_allclose(y, g["projU_y"], rtol=0, atol=1e-4)
    out_names = list(layer.state_dict().keys())
    assert sorted(out_names) == (sorted(names) if not fuse_here else sorted(_proj_shapes(512)))


def test_projection_fold_cache_and_unfolded_call(gpu):
    """The W5*W4 fold is rebuilt when the weights change (load_state_dict / in-place update), and the C entry point
    gives the same answer when the caller passes no fold (w54t = NULL: folded per call into the workspace)."""
    import ctypes as C
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    D = 448
    cfg = R.Config()
    cfg.update(device=gpu)
    layer = R.ProjectionLayer(cfg, D).eval()
    x = torch.from_numpy(synth.rows(0, 9 * 5, D, 31).reshape(9, 5, D)).to(gpu)
    for seed in (11, 12):
        sd = synth.fill_state_dict(_proj_shapes(D), seed)
        layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        with torch.no_grad():
            y = layer(x).cpu().numpy()
        np.testing.assert_allclose(y, O.projection_forward(x.cpu().numpy(), sd), rtol=0, atol=1e-4)
    with torch.no_grad():
        layer.weight_sum.bias.add_(0.25)            # in-place edit: the cached fold must not survive it
        sd["weight_sum.bias"] = sd["weight_sum.bias"] + np.float32(0.25)
        y = layer(x).cpu().numpy()
    np.testing.assert_allclose(y, O.projection_forward(x.cpu().numpy(), sd), rtol=0, atol=1e-4)
    lib = _lib.load()
    w, keep = layer._weights(x.device)
    w.w54t, w.b54 = None, None
    need = lib.radad_projection_workspace_bytes(9, 5, D, 256, 128)
    ws = torch.empty(int(need), dtype=torch.uint8, device=gpu)
    out = torch.empty((9, 128), device=gpu)
    _lib.check(lib.radad_projection_forward(C.byref(w), x.data_ptr(), 9, 5, D, 256, 128, out.data_ptr(), ws.data_ptr(),
                                            int(need), x.device.index, _lib.stream_ptr(x.device)), "forward")
    np.testing.assert_allclose(out.cpu().numpy(), y, rtol=0, atol=2e-6)
    with pytest.raises(ValueError, match="workspace too small"):
        _lib.check(lib.radad_projection_forward(C.byref(w), x.data_ptr(), 9, 5, D, 256, 128, out.data_ptr(), ws.data_ptr(),
                                                16, x.device.index, _lib.stream_ptr(x.device)), "forward")


def test_radad_model_matches_reference_golden(gpu, golden_dir):
    """radad_model.py:32-41 end to end: projection -> fuse Linear over [tpp ; proj] -> detection MLP (BatchNorm in
    eval form), against the reference module's own outputs and, at a ragged batch, against the oracle."""
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    g = np.load(os.path.join(golden_dir, "projection.npz"))
    names = [str(n) for n in g["radad_names"]]
    shapes = {n: tuple(int(v) for v in str(s).split(",") if v) for n, s in zip(names, g["radad_shapes"])}
    sd = synth.fill_state_dict(shapes, int(g["radad_seed"]))
    cfg = R.Config()
    cfg.update(device=gpu)
    D = 512
    model = R.RADADModel(cfg, D).eval()
    assert list(model.state_dict().keys()) == names                                 # the reference's own key names
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x, t = torch.from_numpy(g["radad_x"]).to(gpu), torch.from_numpy(g["radad_t"]).to(gpu)
    with torch.no_grad():
        proj = model.projection_layer(x)
        logits, fused = model.fuse_and_detect(t, proj, return_fused=True)
        logits2 = model(x, t)
    np.testing.assert_allclose(proj.cpu().numpy(), g["radad_proj"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(fused.cpu().numpy(), g["radad_fused"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logits.cpu().numpy(), g["radad_logits"], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(logits.cpu().numpy(), logits2.cpu().numpy())
    assert logits.shape == (3,)
    B = 301                                                                         # several 128-row tiles + a tail
    xb = synth.rows(0, B * 5, D, 401).reshape(B, 5, D)
    tb = synth.rows(0, B, D, 402)
    with torch.no_grad():
        lb = model(torch.from_numpy(xb).to(gpu), torch.from_numpy(tb).to(gpu)).cpu().numpy()
        pb = model.predict_proba(torch.from_numpy(xb).to(gpu), torch.from_numpy(tb).to(gpu)).cpu().numpy()
    _, _, want = O.radad_model_forward(xb, tb, sd)
    np.testing.assert_allclose(lb, want, rtol=0, atol=1e-4)
    np.testing.assert_allclose(pb, 1.0 / (1.0 + np.exp(-want)), rtol=0, atol=1e-4)
    model.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        model(x, t)


@pytest.mark.parametrize("rows,n_out,n_in,act", [(1, 256, 5376, 0), (37, 100, 516, 1), (300, 130, 64, 2), (0, 8, 8, 0)])
def test_linear_forward_split_k(gpu, rows, n_out, n_in, act):
    """radad_linear_forward = nn.Linear (+ tanh / relu) for shapes from one row x wide K (many K-splits) to ragged."""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    lib = _lib.load()
    x = synth.rows(0, max(rows, 1), n_in, 5)[:rows]
    w = synth.rows(0, n_out, n_in, 6) / np.float32(np.sqrt(n_in))
    b = synth.rows(0, 1, n_out, 7)[0]
    xd, wd, bd = (torch.from_numpy(np.ascontiguousarray(a)).to(gpu) for a in (x, w, b))
    out = torch.full((rows, n_out), 7.0, device=gpu)
    need = lib.radad_linear_workspace_bytes(rows, n_out, n_in)
    ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=gpu)
    _lib.check(lib.radad_linear_forward(xd.data_ptr(), n_in, wd.data_ptr(), n_in, bd.data_ptr(), act, rows, n_out, n_in,
                                        out.data_ptr(), n_out, ws.data_ptr(), int(ws.numel()), xd.device.index,
                                        _lib.stream_ptr(xd.device)), "radad_linear_forward")
    want = x.astype(np.float64) @ w.astype(np.float64).T + b
    want = np.tanh(want) if act == 1 else (np.maximum(want, 0) if act == 2 else want)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=1e-4)
    with pytest.raises(ValueError, match="multiples of 4"):
        _lib.check(lib.radad_linear_forward(xd.data_ptr(), n_in, wd.data_ptr(), n_in, bd.data_ptr(), act, 1, n_out, n_in - 1,
                                            out.data_ptr(), n_out, ws.data_ptr(), int(ws.numel()), xd.device.index,
                                            _lib.stream_ptr(xd.device)), "radad_linear_forward")


class _FakeDataset:
    """what process_audio_batch needs from AudioDataset: load_audio(path) -> float32 [N] (dataset.py:139-153)"""

    def __init__(self, clips):
        self.clips = clips

    def load_audio(self, path):
        return self.clips.get(path)


def test_process_audio_batch_and_retrieve(gpu, tmp_path):
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    cfg = R.Config()
    cfg.update(device=gpu, feature_dim=64, tpp_levels=[1, 2, 4], top_k=3, vector_db_index_type="L2",
               vector_db_path=str(tmp_path / "vdb"))
    pipe = R.HotPathPipeline(cfg)
    D = pipe.tpp.get_output_dim()
    assert D == 7 * 64
    # ---- empty store: zeros / NaN with all four arities (pipeline.py:465-476)
    q0 = torch.zeros(2, D, device=gpu)
    v, l = pipe.retrieve_similar_vectors(q0)
    assert v.shape == (2, 3, D) and l.shape == (2, 3) and float(v.abs().sum()) == 0
    v, l, p, d = pipe.retrieve_similar_vectors(q0, return_info=True, return_distances=True)
    assert p == [["", "", ""], ["", "", ""]] and bool(torch.isnan(d).all())
    assert len(pipe.retrieve_similar_vectors(q0, return_info=True)) == 3
    assert len(pipe.retrieve_similar_vectors(q0, return_distances=True)) == 3

    # ---- segment + embed through the reference's signature
    lens = [48000, 48000, 64000, 20000, 70001, 48000]          # the reference's 3.0 s clips, plus ragged ones
    wav = synth.audio(0, len(lens), max(lens), 1234)
    clips = {f"/data/set/clip{i}.wav": wav[i, :n] for i, n in enumerate(lens)}
    ds = _FakeDataset(clips)
    paths = list(clips)
    emb = pipe.process_audio_batch(paths, ds)
    assert emb.shape == (len(lens), D) and emb.is_cuda
    fe = pipe.feature_extractor
    ref = O.embed_clips([clips[p] for p in paths], 32000, 16000, fe.proj_w, fe.proj_b, (1, 2, 4), "max")
    np.testing.assert_allclose(emb.cpu().numpy(), ref, rtol=0, atol=1e-4)
    with pytest.raises(RuntimeError, match="Failed to load"):
        pipe.process_audio_batch(["/nope.wav"], ds)              # pipeline.py:398-399

    # ---- build a store that contains the queries themselves (same basenames) plus distractors
    emb_h = emb.cpu().numpy()
    noise = synth.rows(0, 200, D, 5) * np.float32(0.01 * np.abs(emb_h).mean())
    db = np.concatenate([emb_h, emb_h[np.arange(200) % len(lens)] + noise]).astype(np.float32)
    db_paths = [f"/train/clip{i}.wav" for i in range(len(lens))] + [f"/train/other{i}.wav" for i in range(200)]
    labels = [float(i % 2) for i in range(len(db))]
    pipe.vector_db.add_vectors(db, db_paths, labels, {"speaker_id": ["s"] * len(db)})
    K = cfg.top_k
    for exclude_self in (True, False):
        vec, lbl, rp, dist = pipe.retrieve_similar_vectors(emb, query_paths=paths, exclude_self=exclude_self,
                                                           return_info=True, return_distances=True)
        k_search = K + (10 if exclude_self else 0)                                    # pipeline.py:478
        od, oi = O.knn(db, emb_h, k_search, "L2")
        ov, ol, op, odist = O.retrieve_postprocess(od, oi, db, db_paths, labels, K, D, query_paths=paths,
                                                   exclude_self=exclude_self)
        assert rp == op
        np.testing.assert_array_equal(lbl.cpu().numpy(), ol)
        np.testing.assert_allclose(dist.cpu().numpy(), odist, rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(vec.cpu().numpy(), ov)                           # the STORED rows, bit for bit
        if exclude_self:   # every clipN.wav of the batch is excluded from every row (pipeline.py:463,497-499)
            assert all(os.path.basename(x).startswith("other") for row in rp for x in row)
        else:
            assert [row[0] for row in rp] == [f"/train/clip{i}.wav" for i in range(len(lens))]
    # no query paths: exclusion by training_file_ids (pipeline.py:500-502); too few survivors -> padding (:511-515)
    pipe.training_file_ids = {os.path.basename(p) for p in db_paths[:-2]}
    vec, lbl, rp, dist = pipe.retrieve_similar_vectors(emb[:1], return_info=True, return_distances=True)
    assert all(x == "" or os.path.basename(x) in ("other198.wav", "other199.wav") for x in rp[0])
    n_real = sum(1 for x in rp[0] if x)
    assert bool(torch.isnan(dist[0, n_real:]).all()) and float(vec[0, n_real:].abs().sum()) == 0


def test_full_size_properties(gpu, knn_oracle_lib):
    """BASELINE-sized store (1M x 512, cosine, k=10): size-independent properties + a C-oracle check on a query sample"""
    import torch
    from conftest import c_knn
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import hip_merge, shard_bounds
    lib = _lib.load()
    n, dim, k, nq = 1_000_000, 512, 10, 256
    rows = torch.empty((n, dim), device=gpu)
    _lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, n, dim, 4321, gpu.index or 0, _lib.stream_ptr(gpu)))
    q = torch.empty((nq, dim), device=gpu)
    _lib.check(lib.radad_synth_rows(q.data_ptr(), 0, nq, dim, 977, gpu.index or 0, _lib.stream_ptr(gpu)))
    rows[(torch.arange(nq, device=gpu) * 3907 + 11) % n] = q + 0.05 * rows[:nq]        # one planted neighbour per query
    idx = HipFlatIndex(dim, _lib.METRIC_COSINE, gpu.index or 0)
    idx.add_device(rows)
    D, I, K64 = idx.search_device(q, k, return_f64=True)
    # sortedness, id range/uniqueness, planted neighbour first
    assert bool((D[:, :-1] >= D[:, 1:]).all()) and bool((I >= 0).all()) and bool((I < n).all())
    assert all(len(set(r)) == k for r in I.cpu().tolist())
    assert bool((I[:, 0] == (torch.arange(nq, device=gpu) * 3907 + 11) % n).all())
    # distances equal float64 inner products of the stored rows (reconstruct) with the normalised queries
    qn = torch.empty_like(q)
    _lib.check(lib.radad_rownorm(q.data_ptr(), qn.data_ptr(), nq, dim, gpu.index or 0, _lib.stream_ptr(gpu)))
    rec = idx.reconstruct_batch(I)
    ip = (rec.double() * qn.double()[:, None, :]).sum(-1)
    assert float((ip - K64).abs().max()) < 1e-12 and float((ip.float() - D).abs().max()) < 1e-6
    # idempotence and shard-merge invariance: 4 shards with global ids, merged on the float64 keys
    D2, I2 = idx.search_device(q, k)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    pd, pi = [], []
    for r in range(4):
        lo, hi = shard_bounds(n, 4, r)
        sh = HipFlatIndex(dim, _lib.METRIC_COSINE, gpu.index or 0, id_base=lo)
        sh.add_device(rows[lo:hi])
        _, i_, k_ = sh.search_device(q, k, return_f64=True)
        pd.append(k_); pi.append(i_)
        del sh
    md, mi = hip_merge(_lib.METRIC_COSINE, torch.stack(pd), torch.stack(pi), k)
    assert torch.equal(mi, I) and torch.equal(md, D)
    # C oracle (float64, OpenMP) on a 32-query sample against the rows AS STORED
    stored = torch.empty_like(rows)
    _lib.check(lib.radad_rownorm(rows.data_ptr(), stored.data_ptr(), n, dim, gpu.index or 0, _lib.stream_ptr(gpu)))
    od, oi = c_knn(knn_oracle_lib, stored.cpu().numpy(), qn[:32].cpu().numpy(), k, "IP")
    np.testing.assert_array_equal(I[:32].cpu().numpy(), oi)
    np.testing.assert_allclose(K64[:32].cpu().numpy(), od, rtol=0, atol=1e-12)


def test_build_vector_database_streams_on_device(gpu, tmp_path):
    """pipeline.py:416-447: embed batches and append them to the store without leaving the GPU; then every clip retrieves
    itself first when exclude_self is off"""
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    cfg = R.Config()
    cfg.update(device=gpu, feature_dim=64, tpp_levels=[1, 2, 4], top_k=2, vector_db_index_type="IP", vector_db_path=str(tmp_path / "v"))
    pipe = R.HotPathPipeline(cfg)
    wav = synth.audio(0, 10, 48000, 77)
    clips = {f"/d/c{i}.wav": wav[i] for i in range(10)}
    ds = _FakeDataset(clips)
    paths = list(clips)
    batches = [{"path": paths[:4], "label": torch.tensor([0, 1, 0, 1]), "metadata": {"speaker_id": ["a", "b", "c", "d"]}},
               {"path": paths[4:], "label": [1, 0, 1, 0, 1, 0], "metadata": None}]
    assert pipe.build_vector_database(batches, ds) == 10
    vdb = pipe.vector_db
    assert vdb.index.ntotal == 10 and vdb.vector_paths == paths and vdb.vector_labels == [0, 1, 0, 1, 1, 0, 1, 0, 1, 0]
    assert vdb.vector_metadata["speaker_id"][:5] == ["a", "b", "c", "d", "unknown"]
    assert pipe.training_file_ids == {f"c{i}.wav" for i in range(10)}
    emb = pipe.process_audio_batch(paths, ds)
    vec, lbl, rp = pipe.retrieve_similar_vectors(emb, query_paths=paths, exclude_self=False, return_info=True)
    assert [r[0] for r in rp] == paths
    np.testing.assert_allclose(vec[:, 0].cpu().numpy(), O.maybe_normalize(emb.cpu().numpy(), True), atol=1e-6)
    v2 = R.VectorDatabase(cfg)
    v2.load()                                                   # build_vector_database saved it
    assert v2.index.ntotal == 10


def test_vector_database_sharded_load(gpu, tmp_path):
    """VectorDatabase.load(shard=(rank, world)): two shards of one saved store answer like the whole store after the merge,
    and carry the matching slice of paths / labels."""
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import hip_merge, shard_bounds
    d, n = 64, 1001
    cfg = R.Config()
    cfg.update(device=gpu, vector_db_index_type="IP", vector_db_path=str(tmp_path / "vdb"))
    whole = R.VectorDatabase(cfg)
    whole.create_index(d)
    rows = synth.rows(0, n, d, 515)
    whole.add_vectors(rows, [f"/data/f{i}.wav" for i in range(n)], [i % 2 for i in range(n)], {"i": list(range(n))})
    whole.save()
    q = torch.from_numpy(synth.rows(0, 20, d, 516)).to(gpu)
    qn = torch.nn.functional.normalize(q, dim=1)
    D0, I0, _ = whole.index.search_device(qn, 5, return_f64=True)
    keys, ids = [], []
    for r in range(2):
        part = R.VectorDatabase(cfg)
        part.load(shard=(r, 2))
        lo, hi = shard_bounds(n, 2, r)
        assert part.index.ntotal == hi - lo and part.index.id_base == lo
        assert part.vector_paths == whole.vector_paths[lo:hi] and part.vector_labels == whole.vector_labels[lo:hi]
        assert part.vector_metadata == {"i": list(range(lo, hi))}
        assert part.labels_device().numel() == hi - lo
        _, i, k64 = part.index.search_device(qn, 5, return_f64=True)
        keys.append(k64)
        ids.append(i)
    Dm, Im = hip_merge(_lib.METRIC_COSINE, torch.stack(keys), torch.stack(ids), 5)
    assert torch.equal(Im, I0)
    np.testing.assert_allclose(Dm.cpu().numpy(), D0.cpu().numpy(), rtol=0, atol=1e-6)


def test_projection_layer_extras(gpu):
    """projection.py:125-153 mirrors: get_attention_weights == softmax_K(W2 tanh(W1 x + b1) + b2) in float64;
    memory_efficient_forward == forward; profile_performance runs."""
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    cfg = R.Config()
    cfg.update(device=gpu)
    torch.manual_seed(5)
    layer = R.ProjectionLayer(cfg, 512).eval()
    with torch.no_grad():
        layer.attention_score.bias.normal_(0, 0.1)
        layer.attention_final.bias.normal_(0, 0.1)
    x = torch.randn(70, 5, 512, device=gpu)
    a = layer.get_attention_weights(x)
    xd = x.double()
    s = torch.tanh(xd @ layer.attention_score.weight.double().T + layer.attention_score.bias.double()) @ \
        layer.attention_final.weight.double().T + layer.attention_final.bias.double()
    ref = torch.softmax(s, dim=1)
    assert a.shape == (70, 5, 1) and float((a.double() - ref).abs().max()) < 1e-5
    assert float((a.sum(1) - 1).abs().max()) < 1e-5
    with torch.no_grad():
        full = layer(x)
        assert torch.equal(layer.memory_efficient_forward(x, chunk_size=32), full)
        assert torch.equal(layer.memory_efficient_forward(x[:8], chunk_size=32), full[:8])
    assert layer.profile_performance((4, 5, 512), num_iterations=3) > 0


def test_predict_end_to_end_on_the_reference_store_shape(gpu, tmp_path):
    """pipeline.py:1038-1103 chained: one clip -> embed -> retrieve(exclude_self) -> retry without the exclusion when that leaves
    nothing (:1052-1055) -> RADADModel -> sigmoid, on the reference's Whisper-shaped store (25 423 x 3584: F = 512, levels [1, 2, 4],
    L2, top_k = 5, k_search = 15), against the oracle composition embed_clips -> knn -> retrieve_postprocess -> radad_model_forward.
    Both branches: a store with other files near the query (normal), and a store in which EVERY row carries the query's basename
    (the exclusion removes all 15 hits -> the retry keeps them)."""
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    cfg = R.Config()
    cfg.update(device=gpu, feature_dim=512, tpp_levels=[1, 2, 4], top_k=5, vector_db_index_type="L2",
               vector_db_path=str(tmp_path / "vdb"))
    pipe = R.HotPathPipeline(cfg)
    D = pipe.tpp.get_output_dim()
    assert D == 3584
    wav = synth.audio(0, 1, 48000, 4711)[0]                               # the reference's 3.0 s clip (dataset.py:143)
    qpath = "/eval/query_clip.wav"
    ds = _FakeDataset({qpath: wav})
    fe = pipe.feature_extractor
    emb_ref = O.embed_clips([wav], 32000, 16000, fe.proj_w, fe.proj_b, (1, 2, 4), "max")          # [1, 3584] float64
    n = 25423
    base = synth.rows(0, n, D, 97) * np.float32(0.05 * np.abs(emb_ref).mean()) + emb_ref.astype(np.float32).mean()
    for j in range(40):                                                    # rows near the query: the neighbourhood that gets ranked
        base[(j * 631 + 7) % n] = emb_ref[0].astype(np.float32) + np.float32(0.002 * (j + 1)) * synth.rows(j, 1, D, 98)[0]
    model = R.RADADModel(cfg, D).eval().to(gpu)
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 77)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    for branch in ("normal", "retry"):
        cfg.vector_db_path = str(tmp_path / ("vdb_" + branch))
        pipe.vector_db = R.VectorDatabase(cfg)
        paths = [(f"/train/file{i}.wav" if branch == "normal" else f"/train/dir{i}/query_clip.wav") for i in range(n)]
        labels = [float(i % 2) for i in range(n)]
        pipe.vector_db.add_vectors(base, paths, labels, {"speaker_id": ["s"] * n})
        out = pipe.predict(qpath, ds, model)
        assert pipe.vector_db.index.last_launch()["scan_kind"] in ("hi_smallq", "hi_smallq_ksplit")        # the streaming f16 kernels
        # the oracle's composition, on the embedding the GPU produced (its own parity is test_process_audio_batch_and_retrieve's)
        emb = pipe.process_audio_batch([qpath], ds)
        np.testing.assert_allclose(emb.cpu().numpy(), emb_ref, rtol=0, atol=1e-4)
        eq = emb.cpu().numpy()
        od, oi = O.knn(base, eq, 15, "L2")
        ov, ol, op, _ = O.retrieve_postprocess(od, oi, base, paths, labels, 5, D, query_paths=[qpath], exclude_self=True)
        if branch == "retry":
            assert not ov.any()                                            # every hit carried the query's basename
            od, oi = O.knn(base, eq, 5, "L2")                              # :1053: exclude_self=False searches K, not K + 10
            ov, ol, op, _ = O.retrieve_postprocess(od, oi, base, paths, labels, 5, D, query_paths=[qpath], exclude_self=False)
        _, _, want = O.radad_model_forward(ov, eq, sd)
        assert out["retrieved"] == [{"file": os.path.basename(p), "path": p, "label": int(l)} for p, l in zip(op[0], ol[0])]
        assert out["retrieved_labels"] == [int(l) for l in ol[0]] and out["retrieved_files"] == [os.path.basename(p) for p in op[0]]
        assert abs(out["logit"] - float(want[0])) < 1e-3 * max(1.0, abs(float(want[0])))
        prob = 1.0 / (1.0 + np.exp(-float(want[0])))
        assert abs(out["probability_spoof"] - prob) < 1e-4
        assert out["prediction"] == ("spoof" if prob >= 0.5 else "bona-fide") and set(out) == {
            "prediction", "probability_spoof", "logit", "retrieved_labels", "retrieved_files", "retrieved"}
    # an empty store: zero neighbours, still a verdict (pipeline.py:1039-1040)
    cfg.vector_db_path = str(tmp_path / "vdb_empty")
    pipe.vector_db = R.VectorDatabase(cfg)
    out = pipe.predict(qpath, ds, model)
    assert out["retrieved_files"] == [""] * 5 and out["prediction"] in ("spoof", "bona-fide")
