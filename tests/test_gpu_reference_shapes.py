"""GPU parity at the shapes the REFERENCE itself runs, and the BASELINE configs chained end to end.

Reference shapes (config.py:48-56, pipeline.py:478, vector_database.py:159-182): a store of N = 25 423 clip embeddings of
D = 7 x 768 = 5376 (wav2vec2 / WavLM) or 7 x 512 = 3584 (Whisper), searched with B = 256 (training / evaluation batches) or
B = 1 (predict(), pipeline.py:1038-1054), k = K + 10 = 15, metric L2 (the default vector_db_index_type) or cosine.  These must
take the CERTIFIED f16 kernels (k_knn_hi for batches, k_knn_hi_smallq for the online search) and return the float64 brute
force's ids bit for bit.

Chained configs (BASELINE.json configs 2 and 3): 1024 clips -> embed_clips -> top-10, embeddings against the float64 oracle on
16 clips, neighbour ids against the C oracle on a 64-query sample.
"""
import numpy as np
import pytest

from oracle import radad_oracle as O

pytestmark = pytest.mark.gpu

N_REF = 25423
K_REF = 15


def _dev_rows(gpu, row0, n, dim, seed):
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    t = torch.empty((n, dim), device=gpu, dtype=torch.float32)
    _lib.check(_lib.load().radad_synth_rows(t.data_ptr(), row0, n, dim, seed, gpu.index or 0, _lib.stream_ptr(gpu)))
    return t


def _rownorm(gpu, x):
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    out = torch.empty_like(x)
    _lib.check(_lib.load().radad_rownorm(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], gpu.index or 0, _lib.stream_ptr(gpu)))
    return out


def _np_unit(x):
    x = np.asarray(x, np.float64)
    return x / (np.sqrt((x * x).sum(1))[:, None] + 1e-12)          # vector_database.py:103-104 in float64


@pytest.mark.parametrize("data", ["embedding_like", "random"])
@pytest.mark.parametrize("metric", ["L2", "COSINE"])
@pytest.mark.parametrize("dim", [5376, 3584])
def test_reference_store_shapes_take_the_certified_kernels(gpu, knn_oracle_lib, metric, dim, data):
    import torch
    from conftest import c_knn
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    if data == "random" and dim == 3584:
        pytest.skip("one unstructured configuration per metric is enough")
    n, k, B = N_REF, K_REF, 256
    noise = _dev_rows(gpu, 0, n, dim, 7001)
    qnoise = _dev_rows(gpu, 0, B, dim, 7002)
    if data == "embedding_like":
        # what pooled encoder features look like to a low-precision filter: one large common component (max-pooled activations
        # are positive and share their mean), individual parts an order of magnitude smaller -- every row is within a few
        # 1e-3 of every other in cosine similarity, i.e. within a few eps of the scan
        base = _dev_rows(gpu, 0, 1, dim, 7003).abs() + 0.5
        rows = base + 0.3 * noise
        q = base + 0.3 * qnoise
    else:
        rows, q = noise, qnoise
    jj = torch.arange(B, device=gpu)
    pert = _dev_rows(gpu, 0, 3 * B, dim, 7004)
    for c in range(3):                              # three planted near-duplicates per query: the head of every list is known
        rows[(jj * 97 + c * 7919 + 5) % n] = q + 0.02 * (c + 1) * pert[c * B:(c + 1) * B]
    m = _lib.METRIC_L2 if metric == "L2" else _lib.METRIC_COSINE
    idx = HipFlatIndex(dim, m, gpu.index or 0)
    idx.add_device(rows[:10000])
    idx.add_device(rows[10000:])                    # append path, as add_vectors_batch does (vector_database.py:134-138)
    assert idx.ntotal == n

    # the operands the ranking is defined on: rows as stored, queries as the index normalises them -- and both are checked
    # against an INDEPENDENT float64 computation, so that "as stored" does not lean on the kernels under test
    stored = idx.reconstruct_batch(torch.arange(n, device=gpu)).cpu().numpy()
    rows_h = rows.cpu().numpy()
    if metric == "COSINE":
        np.testing.assert_allclose(stored, _np_unit(rows_h), rtol=0, atol=1e-6)
        qn = _rownorm(gpu, q)
        np.testing.assert_allclose(qn.cpu().numpy(), _np_unit(q.cpu().numpy()), rtol=0, atol=1e-6)
    else:
        np.testing.assert_array_equal(stored, rows_h)                 # L2 stores the rows untouched
        qn = q
    qn_h = qn.cpu().numpy()
    om = "L2" if metric == "L2" else "IP"

    def check(nq, kind, sample):
        D, I, K64 = idx.search_device(q[:nq].contiguous(), k, return_f64=True)
        launch = idx.last_launch()
        assert launch["scan_kind"] == kind, launch
        cert = launch["certificate"]
        assert cert["queries"] == nq and cert["rejected"] <= max(0, nq // 100), cert
        od, oi = c_knn(knn_oracle_lib, stored, qn_h[:sample], k, om)
        assert O.rank_gaps(od).min() > 0, "exact float64 tie in the oracle: pick another seed"
        np.testing.assert_array_equal(I[:sample].cpu().numpy(), oi)
        np.testing.assert_allclose(K64[:sample].cpu().numpy(), od, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(D[:sample].cpu().numpy(), od, rtol=1e-6, atol=1e-6)
        # the planted rows lead every list, in planting order
        want = torch.stack([(jj[:nq] * 97 + c * 7919 + 5) % n for c in range(3)], 1)
        assert torch.equal(I[:, :3], want)
        return I

    I256 = check(B, "hi_tile", 64)                                    # training / evaluation batches (pipeline.py:478-480)
    # embeddings that share most of their mean are scanned on a CENTRED plane (the error bound scales with |y - mean|, not |y|);
    # random directions are not centred; rows of one magnitude share one scale (no per-score arithmetic in the scan)
    plane = idx.plane_info()
    assert plane["built"] and plane["centred"] == (data == "embedding_like") and plane["one_scale"], plane
    I1 = check(1, "hi_smallq", 1)                                     # predict(): one query streams the f16 plane (HBM-bound)
    assert torch.equal(I1[0], I256[0])
    # 16 queries of dim 5376 do not fit the streaming kernel's LDS (172 KB): they take the tile kernel; dim 3584 streams
    I16 = check(16, "hi_tile" if dim == 5376 else "hi_smallq", 16)
    assert torch.equal(I16, I256[:16])
    I13 = check(13, "hi_smallq", 13)                                  # the largest batch that streams at dim 5376
    assert torch.equal(I13, I256[:13])


@pytest.mark.parametrize("reserve", [True, False])
@pytest.mark.parametrize("metric", ["L2", "COSINE"])
def test_a_store_that_drifts_after_the_plane_was_built(gpu, knn_oracle_lib, metric, reserve):
    """The f16 plane's centre mu and scale are decided from the rows the store holds when it is first searched; add_vectors_batch
    appends 10 000 rows at a time (vector_database.py:134-138).  20 k embedding-like rows, a search (the plane is built), then 20 k
    rows around a DIFFERENT common component and 20 k rows 30x larger: ids must stay those of the float64 oracle throughout, and the
    handle must not end on the fp32 fallback -- the plane is decided again (radad_knn_plane_rebuilds) and the search after that runs
    the certified tile scan with <= 2 % of the queries rejected."""
    import torch
    from conftest import c_knn
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    dim, k, B, n1 = 1024, K_REF, 256, 20000
    m = _lib.METRIC_L2 if metric == "L2" else _lib.METRIC_COSINE
    base_a = _dev_rows(gpu, 0, 1, dim, 7101).abs() + 0.5
    base_b = _dev_rows(gpu, 1, 1, dim, 7101).abs() * 2.0 + 0.1                # another encoder's mean
    part = [base_a + 0.3 * _dev_rows(gpu, 0, n1, dim, 7102),
            base_b + 0.3 * _dev_rows(gpu, n1, n1, dim, 7102),
            30.0 * (base_a + 0.3 * _dev_rows(gpu, 2 * n1, n1, dim, 7102))]    # the same family, 30x the magnitude
    idx = HipFlatIndex(dim, m, gpu.index or 0)
    if reserve:
        idx.reserve(3 * n1)            # no reallocation (which drops the plane anyway): the plane must be RE-decided in place
    om = "L2" if metric == "L2" else "IP"
    rows_all = []
    for step, rows in enumerate(part):
        q = rows[torch.arange(B, device=gpu) * 71 % n1] + 0.05 * _dev_rows(gpu, 0, B, dim, 7110 + step)      # queries near this part's rows
        for lo in range(0, n1, 10000):
            idx.add_device(rows[lo:lo + 10000].contiguous())
        rows_all.append(rows)
        stored = idx.reconstruct_batch(torch.arange(idx.ntotal, device=gpu)).cpu().numpy()
        qn_h = (_rownorm(gpu, q) if metric == "COSINE" else q).cpu().numpy()
        od, oi = c_knn(knn_oracle_lib, stored, qn_h[:48], k, om)
        for rep in range(6):                                                   # (the rejection counters lag two searches; the handle may
                                                                               #  first re-decide the plane, then widen its candidate buffers)
            D, I = idx.search_device(q, k)
            launch = idx.last_launch()
            if O.rank_gaps(od).min() > 0:
                np.testing.assert_array_equal(I[:48].cpu().numpy(), oi)
        info = idx.plane_info()
        assert launch["scan_kind"] == "hi_tile", (step, launch, info)          # not on the fp32 fallback
        assert launch["certificate"]["rejected"] <= max(1, B // 50), (step, launch["certificate"], info)
        assert info["built"], info
    if reserve:
        assert idx.plane_info()["rebuilds"] >= 1, idx.plane_info()             # 20 k -> 40 k rows doubled the store: decided again


def _bench_like_store(gpu, emb, n_total, dim, seed=4321, noise_seed=99):
    """the store bench.py builds: synthetic rows + two near-duplicates of every query embedding"""
    import torch
    rows = _dev_rows(gpu, 0, n_total, dim, seed)
    Q = emb.shape[0]
    noise = _dev_rows(gpu, 0, 2 * Q, dim, noise_seed)
    jj = torch.arange(Q, device=gpu)
    scale = emb.norm(dim=1, keepdim=True) / (dim ** 0.5)
    for c, e in ((0, 0.05), (1, 0.10)):
        rows[(jj * 977 + c * 350003 + 17) % n_total] = emb + e * scale * noise[c * Q:(c + 1) * Q]
    return rows


def _chain_check(gpu, knn_oracle_lib, fe, wave, offs_host, emb, n_total):
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    from conftest import c_knn
    B, dim, k = emb.shape[0], emb.shape[1], 10
    # embeddings: the float64 oracle on 16 clips spread over the batch (bar 1e-4, north_star)
    pick = np.linspace(0, B - 1, 16).astype(int)
    wav_h = [wave[int(offs_host[b]):int(offs_host[b + 1])].cpu().numpy() for b in pick]
    ref = O.embed_clips(wav_h, fe.segment_length, fe.hop_length, fe.proj_w, fe.proj_b, (1,), "max")
    np.testing.assert_allclose(emb[torch.from_numpy(pick).to(gpu)].cpu().numpy(), ref, rtol=0, atol=1e-4)
    # retrieve: cosine top-10 against the store, through the reference's call surface
    cfg = R.Config()
    cfg.update(device=gpu, tpp_levels=[1], feature_dim=dim, vector_db_index_type="IP")
    vdb = R.VectorDatabase(cfg)
    vdb.create_index(dim)
    rows = _bench_like_store(gpu, emb, n_total, dim)
    vdb.index.add_device(rows)
    D, I = vdb.search_batch(emb, k=k)
    launch = vdb.index.last_launch()
    assert launch["scan_kind"] == "hi_tile" and launch["certificate"]["rejected"] <= B // 100, launch
    jj = torch.arange(B, device=gpu)
    # (two clips can embed so closely that one's planted row beats the other's own: the oracle below decides; almost all lead)
    assert float((I[:, 0] == (jj * 977 + 17) % n_total).float().mean()) > 0.98
    # independent check of the rows as stored (numpy float64 normalisation of the inputs), then the C oracle on 64 queries
    stored = np.empty((n_total, dim), np.float32)
    for r0 in range(0, n_total, 1 << 17):
        ids = torch.arange(r0, min(n_total, r0 + (1 << 17)), device=gpu)
        stored[r0:r0 + len(ids)] = vdb.index.reconstruct_batch(ids).cpu().numpy()
        np.testing.assert_allclose(stored[r0:r0 + len(ids)], _np_unit(rows[r0:r0 + len(ids)].cpu().numpy()), rtol=0, atol=1e-6)
    sample = torch.from_numpy(np.linspace(0, B - 1, 64).astype(np.int64)).to(gpu)
    qn = _rownorm(gpu, emb[sample].contiguous())
    np.testing.assert_allclose(qn.cpu().numpy(), _np_unit(emb[sample].cpu().numpy()), rtol=0, atol=1e-6)
    od, oi = c_knn(knn_oracle_lib, stored, qn.cpu().numpy(), k, "IP")
    np.testing.assert_array_equal(I[sample].cpu().numpy(), oi)
    np.testing.assert_allclose(D[sample].cpu().numpy(), od, rtol=0, atol=1e-4)


def _extractor(gpu):
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    cfg = R.Config()
    cfg.update(device=gpu, tpp_levels=[1], tpp_pooling_type="max", feature_dim=512, vector_db_index_type="IP")
    return R.MelProjectionFeatureExtractor(cfg)


def test_config2_chained_1k_clips_100k_store(gpu, knn_oracle_lib):
    """BASELINE config 2: 1 k fixed-length 4 s clips -> fp32 embed -> brute-force cosine top-10 against 100 k x 512"""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    fe = _extractor(gpu)
    B, n = 1024, 64000
    wave = torch.empty(B * n, device=gpu, dtype=torch.float32)
    _lib.check(_lib.load().radad_synth_audio(wave.data_ptr(), 0, B, n, 1234, gpu.index or 0, _lib.stream_ptr(gpu)))
    offs = np.arange(B + 1, dtype=np.int64) * n
    emb = fe.embed_clips(wave, offs)
    _chain_check(gpu, knn_oracle_lib, fe, wave, offs, emb, 100_000)


def test_config3_chained_ragged_clips_1m_store(gpu, knn_oracle_lib):
    """BASELINE config 3: 1 k release_in_the_wild-shaped (log-normal, variable-length) clips cut by the segmenter rule, clip
    offsets resident on the DEVICE (plan built by k_build_plan) -> 1 M x 512 store"""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    fe = _extractor(gpu)
    B = 1024
    rng = np.random.default_rng(1235)
    lens = (np.clip(np.exp(rng.normal(np.log(3.6), 0.6, B)), 0.5, 20.0) * 16000).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    # every clip is generated at 20 s with its own seed stream (own sinusoid) and cut to its length, as bench.py --workload ragged does
    wave = torch.empty(int(offs[-1]), device=gpu, dtype=torch.float32)
    full = torch.empty(320000, device=gpu, dtype=torch.float32)
    for b in range(B):
        _lib.check(_lib.load().radad_synth_audio(full.data_ptr(), b, 1, 320000, 1235, gpu.index or 0, _lib.stream_ptr(gpu)))
        wave[int(offs[b]):int(offs[b + 1])] = full[:int(lens[b])]
    emb = fe.embed_clips(wave, torch.from_numpy(offs).to(gpu))
    fe.check_device_plan()
    emb_host_offsets = fe.embed_clips(wave, offs)
    assert torch.equal(emb, emb_host_offsets)                          # device-built plan == host-built plan, bit for bit
    _chain_check(gpu, knn_oracle_lib, fe, wave, offs, emb, 1_000_000)


def test_stored_rows_at_1m_match_an_independent_normalisation(gpu):
    """the 'rows as stored' that the retrieval tests rank on, checked at the headline size against numpy: x / (|x| + 1e-12) in
    float64 of the rows handed to add (vector_database.py:103-104)"""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    n, dim = 1_000_000, 512
    rows = _dev_rows(gpu, 0, n, dim, 4321)
    rows[::1000] *= 1e-3                      # a few rows of very different magnitude
    rows[7] = 0                               # the all-zero row: 0 / (0 + 1e-12) = 0
    idx = HipFlatIndex(dim, _lib.METRIC_COSINE, gpu.index or 0)
    idx.add_device(rows)
    worst = 0.0
    for r0 in range(0, n, 1 << 17):
        ids = torch.arange(r0, min(n, r0 + (1 << 17)), device=gpu)
        got = idx.reconstruct_batch(ids).cpu().numpy()
        worst = max(worst, float(np.abs(got - _np_unit(rows[r0:r0 + len(ids)].cpu().numpy())).max()))
    assert worst < 1e-6, worst
    assert float(idx.reconstruct_batch(torch.tensor([7], device=gpu)).abs().max()) == 0.0


@pytest.mark.parametrize("kind", ["flat_small", "flat_tile", "ivf"])
def test_non_finite_queries_are_contained(gpu, kind):
    """a NaN / inf query must not leave output slots unwritten nor index anything out of range: its row comes back as faiss
    fills what it cannot find (-1), every other query of the batch is answered as usual"""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, HipIVFFlatIndex, _lib
    dim, k = 128, 10
    n = 20000 if kind != "flat_small" else 3000
    rows = _dev_rows(gpu, 0, n, dim, 8101)
    nq = 40 if kind != "flat_small" else 5
    q = _dev_rows(gpu, 0, nq, dim, 8102)
    if kind == "ivf":
        idx = HipIVFFlatIndex(dim, 64, gpu.index or 0)
        idx.train(rows[:5000])
        idx.add(rows)
        idx.nprobe = 8
    else:
        idx = HipFlatIndex(dim, _lib.METRIC_L2, gpu.index or 0)
        idx.add_device(rows)
    D0, I0 = idx.search_device(q, k)
    bad = q.clone()
    bad[1, 3] = float("nan")
    bad[2, :] = float("nan")
    bad[3, 5] = float("inf")
    D, I = idx.search_device(bad, k)
    torch.cuda.synchronize()
    ok = [i for i in range(nq) if i not in (1, 2, 3)]
    assert torch.equal(I[ok], I0[ok]) and torch.equal(D[ok], D0[ok])
    for r in (1, 2, 3):
        ids = I[r].cpu().numpy()
        assert ((ids >= -1) & (ids < n)).all(), ids                    # nothing uninitialised, nothing out of range
        real = ids[ids >= 0]
        assert len(set(real.tolist())) == len(real)
    assert (I[2] == -1).all()


def test_device_offsets_outside_the_wave_buffer_are_clamped_and_reported(gpu):
    """radad_embed_forward_dev cannot validate device-resident offsets on the host; k_build_plan clamps them (no read outside
    the wave buffer, no scratch overrun) and the repair is reported by check_device_plan / the next call"""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    fe = _extractor(gpu)
    n = 64000
    wave = torch.empty(4 * n, device=gpu, dtype=torch.float32)
    _lib.check(_lib.load().radad_synth_audio(wave.data_ptr(), 0, 4, n, 1234, gpu.index or 0, _lib.stream_ptr(gpu)))
    good = torch.arange(5, device=gpu, dtype=torch.int64) * n
    ref = fe.embed_clips(wave, good)
    fe.check_device_plan()                                             # fine: no exception
    for offs, why in (([0, n, 2 * n, 3 * n, 40 * n], "outside the wave buffer"),           # last clip runs far past the end
                      ([0, n, 2 * n, 3 * n, 2 ** 40], "outside the wave buffer"),
                      ([-5 * n, n, 2 * n, 3 * n, 4 * n], "outside the wave buffer"),
                      ([0, 2 * n, n, 3 * n, 4 * n], "not non-decreasing")):
        out = fe.embed_clips(wave, torch.tensor(offs, device=gpu, dtype=torch.int64))
        torch.cuda.synchronize()                                       # a fault would surface here
        with pytest.raises(ValueError, match=why):
            fe.check_device_plan()
        # clips whose own offsets were fine are embedded as before (clip 1 in all four cases)
        if why == "outside the wave buffer":
            assert torch.equal(out[1], ref[1])
    # the flag is reported once; the next good batch is clean, and a bad batch that has COMPLETED is also reported by the next call
    # (which polls and never waits: ADVICE r3 -- it used to block behind the previous batch)
    bad = torch.tensor([0, n, 2 * n, 3 * n, 40 * n], device=gpu, dtype=torch.int64)
    fe.embed_clips(wave, bad)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="an earlier device-offset batch") as ei:
        fe.embed_clips(wave, good)
    assert torch.equal(ei.value.result, ref)                           # the valid batch of the raising call WAS enqueued: nothing is lost
    out = fe.embed_clips(wave, good)
    fe.check_device_plan()
    assert torch.equal(out, ref)
    # six batches queued back to back (more than the ring of report words holds), the first one bad: nothing is lost, the report
    # comes once -- from a poll of a later call if that batch happens to be complete by then, from the final check otherwise
    seen = 0
    for i in range(6):
        try:
            fe.embed_clips(wave, bad if i == 0 else good)
        except ValueError:
            seen += 1
    try:
        fe.check_device_plan()
    except ValueError:
        seen += 1
    assert seen == 1
    fe.check_device_plan()                                             # reported once


def test_kth_largest_kernel(gpu):
    """radad_kth_largest (the sharded search's global bound: k-th largest of the G k lower bounds of a query) == torch.topk"""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex
    g = torch.Generator(device="cpu").manual_seed(5)
    for G, nq, k in ((8, 1000, 10), (2, 33, 15), (8, 50, 128), (5, 7, 1)):
        x = torch.randn((G, nq, k), generator=g)
        x[0, 0, :] = float("-inf")                       # a shard with nothing to offer
        x[1, 1, 0] = float("nan")                        # ranks lowest
        if k > 1:
            x[:, 2, :] = 0.25                            # all equal
        want = torch.topk(torch.nan_to_num(x, nan=float("-inf")).permute(1, 0, 2).reshape(nq, G * k), k, dim=1).values[:, k - 1]
        got = HipFlatIndex.global_bound(x.to(gpu), k).cpu()
        assert torch.equal(got, want), (G, nq, k)


def test_certificate_reports_are_acted_on_while_the_host_runs_ahead(gpu):
    """the handle's self-tuning reads the certificate's reports from pinned memory without waiting.  Round 4 acted only on
    the report of exactly the search two back -- a host that queues searches faster than the device runs them (a bench loop, back to
    back predict() calls) never saw one and a mostly-rejected store stayed on the exact kernel for every query.  Here: a store whose
    rows are within 2 eps of each other cluster by cluster (every query overflows its candidate buffer), searches queued WITHOUT any host
    synchronisation: the handle must retune (wider candidate buffers, then the fp32 kernels) while the queue is running, and the
    results stay those of the float64 brute force."""
    import torch
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    n, dim, B, k = 40000, 128, 256, 10
    base = _dev_rows(gpu, 0, 1, dim, 9301)
    # two tight clusters at +base and -base: the column mean is ~0, so the plane is NOT centred and the bound scales with |y| = |base|,
    # while the 20 000 rows of a query's own cluster differ by 1e-4: all of them are within 2 eps of its k-th best
    sign = (torch.arange(n, device=gpu) % 2).float()[:, None] * 2.0 - 1.0
    rows = sign * base + 1e-4 * _dev_rows(gpu, 0, n, dim, 9302)
    q = base + 0.5 * _dev_rows(gpu, 0, B, dim, 9303)
    idx = HipFlatIndex(dim, _lib.METRIC_L2, gpu.index or 0)
    idx.add_device(rows)
    torch.cuda.synchronize()
    retuned_at = None
    for i in range(400):
        D, I = idx.search_device(q, k)                     # (asynchronous: nothing here waits for the device)
        t = idx.tuning_info()
        if t["cap_boost"] > 1 or t["fp32_searches_left"] > 0:
            retuned_at = i
            break
    assert retuned_at is not None and t["reports_consumed"] >= 1, t
    torch.cuda.synchronize()
    D, I = idx.search_device(q, k)
    od, oi = O.knn(rows.cpu().numpy(), q.cpu().numpy()[:16], k, "L2")
    if O.rank_gaps(od).min() > 0:
        np.testing.assert_array_equal(I[:16].cpu().numpy(), oi)
    np.testing.assert_allclose(D[:16].cpu().numpy(), od, rtol=1e-5, atol=1e-7)
