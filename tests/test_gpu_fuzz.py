"""Bounded randomised parity sweeps (collected by pytest, ~30 s on the GPU box; seeds fixed so a failure reproduces).
kNN: random store sizes / batch sizes / dims / k / metrics / store dtypes / append patterns / id bases against the float64
C oracle (oracle/knn_oracle.c) over the rows AS STORED.  Embedding: random segment lengths / overlaps / clip lengths /
feature dims / pyramid levels / pooling modes / amplitudes against the numpy float64 oracle, through both the host-offset
and the device-offset entry points."""
import numpy as np
import pytest

from oracle import radad_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_knn(gpu, knn_oracle_lib, seed):
    import torch
    from conftest import c_knn
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    rng = np.random.default_rng(seed)
    for case in range(14):
        metric = ["L2", "IP", "COSINE"][rng.integers(3)]
        dim = int(rng.choice([32, 64, 96, 128, 256, 512, 100, 36]))
        n = int(rng.choice([1, 17, 255, 256, 257, 1000, 4097, 20000, 70001]))
        nq = int(rng.choice([1, 16, 17, 33, 128, 129, 255, 256, 257, 600]))
        k = int(rng.choice([1, 5, 10, 11, 15, 24, 26, 40]))
        f16 = bool(rng.integers(4) == 0)
        id_base = int(rng.choice([0, 0, 12345678901]))
        db = synth.rows(0, n, dim, 9000 + 100 * seed + case)
        q = synth.rows(0, nq, dim, 9500 + 100 * seed + case)
        if rng.integers(2) and not f16 and metric != "COSINE":
            db *= np.exp2(rng.integers(-10, 10, size=n)).astype(np.float32)[:, None]
        for j in range(min(nq, 50)):
            db[(j * 31 + 7) % n] = q[j] + np.float32(0.05) * synth.rows(j, 1, dim, 9900 + case)[0]
        m = {"L2": _lib.METRIC_L2, "IP": _lib.METRIC_IP, "COSINE": _lib.METRIC_COSINE}[metric]
        idx = HipFlatIndex(dim, m, 0, id_base, store_f16=f16)
        cut = int(rng.integers(0, n + 1))
        if cut:
            idx.add(db[:cut])
        if cut and rng.integers(2):
            idx.search(q[: min(nq, 300)], min(k, cut))       # a search between the appends
        if cut < n:
            idx.add(db[cut:])
        D, I = idx.search(q, k)
        rec = idx.reconstruct_batch(torch.arange(id_base, id_base + n, device=gpu)).cpu().numpy()
        qq = q.astype(np.float64)
        if metric == "COSINE":
            qq = qq / (np.sqrt((qq ** 2).sum(1))[:, None] + 1e-300)
        kk = min(k, n)
        od, oi = c_knn(knn_oracle_lib, rec, qq.astype(np.float32), kk, "L2" if metric == "L2" else "IP", id_base)
        ok = np.array_equal(I[:, :kk], oi)
        if not ok and metric == "COSINE":
            # the C oracle saw float32-rounded normalised queries; ranks within 1e-8 may differ: compare on float64 scores
            sc = rec.astype(np.float64) @ qq.T
            ok = all(np.allclose(np.sort(sc[I[j, :kk] - id_base, j])[::-1], np.sort(sc[oi[j] - id_base, j])[::-1], rtol=0, atol=1e-7)
                     for j in range(nq))
        info = idx.last_launch()
        assert ok and np.all(I[:, kk:] == -1), dict(seed=seed, case=case, metric=metric, dim=dim, n=n, nq=nq, k=k, f16=f16,
                                                    id_base=id_base, info=info)


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_embed(gpu, seed):
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    rng = np.random.default_rng(seed)
    for case in range(10):
        seg_s = float(rng.choice([0.03, 0.1, 0.5, 1.0, 1.37, 2.0]))
        L = int(round(seg_s * 100) / 100 * 16000) // 160 * 160
        seg_s = L / 16000
        overlap = float(rng.choice([0.0, 0.25, 0.5, 0.75]))
        F = int(rng.choice([32, 64, 96, 256, 512]))
        levels = [[1], [1, 2], [1, 2, 4], [1, 3, 5], [2]][rng.integers(5)]
        mode = ["max", "avg"][rng.integers(2)]
        norm = bool(rng.integers(2))
        gain = float(rng.choice([1.0, 1e-4, 3e3]))
        cfg = R.Config()
        cfg.update(device=gpu, feature_dim=F, tpp_levels=levels, tpp_pooling_type=mode, segment_length=seg_s, segment_overlap=overlap,
                   melproj_normalize=norm, melproj_seed=100 + 10 * seed + case)
        fe = R.MelProjectionFeatureExtractor(cfg)
        nclip = int(rng.integers(1, 6))
        lens = [int(x) for x in rng.choice([1, L // 3 + 1, L - 1, L, L + 1, 2 * L + 7, 3 * fe.hop_length + L, 50000], size=nclip)]
        wav = synth.audio(0, nclip, max(lens), 7000 + 100 * seed + case) * np.float32(gain)
        clips = [wav[i, :n] for i, n in enumerate(lens)]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        wave = torch.from_numpy(np.concatenate(clips)).to(gpu)
        emb = fe.embed_clips(wave, offs)
        emb_dev = fe.embed_clips(wave, torch.from_numpy(offs).to(gpu))          # segment plan built on the device
        ref = O.embed_clips(clips, fe.segment_length, fe.hop_length, fe.proj_w, fe.proj_b, tuple(levels), mode, normalize=norm)
        err = float(np.abs(emb.cpu().numpy() - ref).max())
        what = dict(seed=seed, case=case, L=L, hop=fe.hop_length, F=F, levels=levels, mode=mode, norm=norm, gain=gain, lens=lens, err=err)
        assert emb.shape == ref.shape and err < 1e-4, what
        assert torch.equal(emb, emb_dev), what


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_scan_phases_and_planes(gpu, knn_oracle_lib, seed):
    """the round-3 machinery of the certified tile scan under random shapes: stores below / at / above the phase boundaries (one
    launch, two, three), batches of 17 .. 2100 queries (1 .. 9 query tiles: both XCD mappings), k from 1 to 128, rows with and
    without a common component (centred / un-centred plane) and of one or of many magnitudes (one scale / per-row scales), appends
    after the plane was built.  Rows are generated on the device; ids against the C oracle on a 24-query sample of the rows AS
    STORED."""
    import torch
    from conftest import c_knn
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    lib = _lib.load()
    rng = np.random.default_rng(seed)

    def dev_rows(row0, n, dim, sd):
        t = torch.empty((n, dim), device=gpu, dtype=torch.float32)
        _lib.check(lib.radad_synth_rows(t.data_ptr(), row0, n, dim, sd, gpu.index or 0, _lib.stream_ptr(gpu)))
        return t

    for case in range(6):
        metric = ["L2", "IP", "COSINE"][rng.integers(3)]
        dim = int(rng.choice([64, 128, 256, 512]))
        n = int(rng.choice([16384, 20000, 131072, 131072 + 300, 200000, 1_200_000 if dim <= 128 else 400000]))
        nq = int(rng.choice([17, 255, 256, 257, 1000, 2100]))
        k = int(rng.choice([1, 10, 15, 26, 27, 64, 128]))
        common = bool(rng.integers(2))
        ragged_scale = bool(rng.integers(3) == 0) and metric != "COSINE"
        rows = dev_rows(0, n, dim, 20000 + 100 * seed + case)
        q = dev_rows(0, nq, dim, 21000 + 100 * seed + case)
        if common:
            base = dev_rows(0, 1, dim, 22000 + case).abs() + 0.5
            rows = base + 0.3 * rows
            q = base + 0.3 * q
        if ragged_scale:
            rows *= torch.exp2(torch.arange(n, device=gpu) % 19 - 9).float()[:, None]
        jj = torch.arange(nq, device=gpu)
        rows[(jj * 131 + 7) % n] = q + 0.03 * dev_rows(0, nq, dim, 23000 + case)       # one planted neighbour per query
        m = {"L2": _lib.METRIC_L2, "IP": _lib.METRIC_IP, "COSINE": _lib.METRIC_COSINE}[metric]
        idx = HipFlatIndex(dim, m, gpu.index or 0)
        cut = int(rng.choice([n, n, n // 2, n - 1000]))
        idx.add_device(rows[:cut])
        if cut < n:
            idx.search_device(q[:64].contiguous(), min(k, 10))                        # builds the plane on the first part
            idx.add_device(rows[cut:])
        D, I, K64 = idx.search_device(q, k, return_f64=True)
        info, plane = idx.last_launch(), idx.plane_info()
        what = dict(seed=seed, case=case, metric=metric, dim=dim, n=n, nq=nq, k=k, common=common, ragged_scale=ragged_scale, cut=cut,
                    plane=plane, info=info)
        # (the floor's rank k + 6 must exist twice over in the sample -- 16 entries per sample tile, at most 64 tiles and an eighth
        # of the store: a small store with a large k takes the fp32 kernels by design)
        sample_entries = min(64, n // 2048) // 8 * 8 * 16
        assert info["scan_kind"] == ("hi_tile" if sample_entries >= 2 * (k + 6) else "f32_tile"), what
        if info["scan_kind"] != "hi_tile":
            continue
        if not ragged_scale:
            assert info["certificate"]["rejected"] <= max(1, nq // 50), what
        assert bool((I[:, 0] == (jj * 131 + 7) % n).all()) or metric == "IP", what     # (raw inner product favours long rows)
        sample = torch.from_numpy(rng.choice(nq, size=min(24, nq), replace=False)).to(gpu)
        stored = idx.reconstruct_batch(torch.arange(n, device=gpu)).cpu().numpy()
        qs = q[sample].contiguous()
        if metric == "COSINE":
            qn = torch.empty_like(qs)
            _lib.check(lib.radad_rownorm(qs.data_ptr(), qn.data_ptr(), qs.shape[0], dim, gpu.index or 0, _lib.stream_ptr(gpu)))
            qs = qn
        od, oi = c_knn(knn_oracle_lib, stored, qs.cpu().numpy(), k, "L2" if metric == "L2" else "IP")
        gaps_ok = O.rank_gaps(od).min() > 0
        I_s = I[sample].cpu().numpy()
        if gaps_ok:
            assert np.array_equal(I_s, oi), what
        np.testing.assert_allclose(K64[sample].cpu().numpy(), od, rtol=1e-9, atol=1e-9, err_msg=str(what))
        del idx, rows, stored


@pytest.mark.parametrize("kernel", ["fft", "gemm"])
@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_embed_shared_frames(gpu, seed, kernel):
    """the shared-frame log-mel kernels (k_logmel_fft_clip: radix FFT on the vector ALU; k_logmel_h_clip: DFT-as-GEMM on the matrix
    pipe) under random segment shapes: T = 8 .. 200 frames per segment, segment hops
    of 2 .. 150 frames with one to four owners per frame, clips from shorter than a segment (zero padded) to dozens of segments
    (edge-only chunks), all pooling options; against the numpy float64 oracle, host- and device-resident offsets bit for bit."""
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    rng = np.random.default_rng(seed)
    shapes = [(8, 2), (8, 4), (10, 5), (24, 8), (40, 10), (40, 30), (100, 25), (100, 50), (100, 75), (200, 50), (200, 100), (200, 150)]
    for case in range(8):
        T, H = shapes[rng.integers(len(shapes))]
        L, hop = 160 * T, 160 * H
        F = int(rng.choice([32, 64, 256]))
        levels = [[1], [1, 2], [1, 2, 4], [3]][rng.integers(4)]
        mode = ["max", "avg"][rng.integers(2)]
        norm = bool(rng.integers(4) != 0)
        cfg = R.Config()
        cfg.update(device=gpu, feature_dim=F, tpp_levels=levels, tpp_pooling_type=mode, segment_length=L / 16000,
                   segment_overlap=1.0 - (hop + 0.25) / L,               # (the mirror takes int(L (1 - overlap)) like the reference)
                   melproj_normalize=norm, melproj_seed=300 + 10 * seed + case, melproj_logmel_fft=(kernel == "fft"))
        fe = R.MelProjectionFeatureExtractor(cfg)
        assert (fe.segment_length, fe.hop_length) == (L, hop), (T, H, fe.segment_length, fe.hop_length)
        nclip = int(rng.integers(1, 5))
        budget = 150000                                                   # samples per clip at most (the oracle is numpy float64)
        lens = [int(x) for x in rng.choice([L // 2 + 3, L, L + 1, L + hop, L + 3 * hop + 77, min(budget, L + 40 * hop + 5)], size=nclip)]
        gain = np.float32(rng.choice([1.0, 1e-3, 50.0]))
        wav = synth.audio(0, nclip, max(lens), 8000 + 100 * seed + case) * gain
        if rng.integers(2):
            # a DC offset of the signal's own order: the mean correction of bands 0 and 1.  (Not 0.3 beside an amplitude of 1e-3: the
            # segment statistics are float32 two-pass sums, as HF's float32 numpy arrays give the reference -- with the mean 1500 x the
            # standard deviation that alone moves an embedding by 5.6e-4 against this float64 oracle, in the per-segment kernel too.)
            wav += np.float32(0.3) * gain
        clips = [wav[i, :n] for i, n in enumerate(lens)]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        wave = torch.from_numpy(np.concatenate(clips)).to(gpu)
        emb = fe.embed_clips(wave, offs)
        what = dict(seed=seed, case=case, T=T, H=H, F=F, levels=levels, mode=mode, norm=norm, lens=lens)
        assert fe.last_logmel_kind() == ("clip_frames_fft" if kernel == "fft" else "clip_frames"), what
        emb_dev = fe.embed_clips(wave, torch.from_numpy(offs).to(gpu))
        ref = O.embed_clips(clips, L, hop, fe.proj_w, fe.proj_b, tuple(levels), mode, normalize=norm)
        err = float(np.abs(emb.cpu().numpy() - ref).max())
        assert emb.shape == ref.shape and err < 1e-4, dict(what, err=err)
        assert torch.equal(emb, emb_dev), what


@pytest.mark.parametrize("seed", [31, 32])
def test_fuzz_small_batches_and_small_stores(gpu, knn_oracle_lib, seed):
    """the chains round 4 added, under random shapes: batches of 1..16 queries (streaming scans with 16- or 32-entry lists,
    k_refine_small with 8 workgroups per query, the K-split form on wide rows) and stores of a few thousand rows (k_knn_dense), all
    metrics, embedding-like rows with a common component or not, rows of many magnitudes; ids and float64 keys against the C oracle
    on the rows AS STORED."""
    import torch
    from conftest import c_knn
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib
    lib = _lib.load()
    rng = np.random.default_rng(seed)

    def dev_rows(row0, n, dim, sd):
        t = torch.empty((n, dim), device=gpu, dtype=torch.float32)
        _lib.check(lib.radad_synth_rows(t.data_ptr(), row0, n, dim, sd, gpu.index or 0, _lib.stream_ptr(gpu)))
        return t

    for case in range(8):
        metric = ["L2", "IP", "COSINE"][rng.integers(3)]
        n, dim = [(3000, 256), (6144, 64), (16384, 128), (20000, 512), (70000, 128), (25423, 1024), (9000, 2048)][rng.integers(7)]
        nq = int(rng.choice([1, 2, 7, 16] if n > 6144 else [1, 16, 40, 700]))
        k = int(rng.choice([1, 10, 15, 16, 26]))
        common = bool(rng.integers(2))
        ragged_scale = bool(rng.integers(3) == 0) and metric != "COSINE"
        rows = dev_rows(0, n, dim, 30000 + 100 * seed + case)
        q = dev_rows(0, nq, dim, 31000 + 100 * seed + case)
        if common:
            base = dev_rows(0, 1, dim, 32000 + case).abs() + 0.5
            rows = base + 0.3 * rows
            q = base + 0.3 * q
        if ragged_scale:
            rows *= torch.exp2(torch.arange(n, device=gpu) % 19 - 9).float()[:, None]
        jj = torch.arange(nq, device=gpu)
        rows[(jj * 131 + 7) % n] = q + 0.03 * dev_rows(0, nq, dim, 33000 + case)
        m = {"L2": _lib.METRIC_L2, "IP": _lib.METRIC_IP, "COSINE": _lib.METRIC_COSINE}[metric]
        idx = HipFlatIndex(dim, m, gpu.index or 0)
        idx.add_device(rows)
        D, I, K64 = idx.search_device(q, k, return_f64=True)
        info = idx.last_launch()
        what = dict(seed=seed, case=case, metric=metric, dim=dim, n=n, nq=nq, k=k, common=common, ragged_scale=ragged_scale, info=info,
                    plane=idx.plane_info())
        if n <= 6144:
            assert info["scan_kind"] == "f32_dense", what
        elif n >= 16384 and dim % 64 == 0:
            assert info["scan_kind"] == "hi_smallq", what
        stored = idx.reconstruct_batch(torch.arange(n, device=gpu)).cpu().numpy()
        qs = q.contiguous()
        if metric == "COSINE":
            qn = torch.empty_like(qs)
            _lib.check(lib.radad_rownorm(qs.data_ptr(), qn.data_ptr(), qs.shape[0], dim, gpu.index or 0, _lib.stream_ptr(gpu)))
            qs = qn
        sel = np.arange(nq)[:24]
        od, oi = c_knn(knn_oracle_lib, stored, qs[:24].cpu().numpy(), k, "L2" if metric == "L2" else "IP")
        if O.rank_gaps(od).min() > 0:
            assert np.array_equal(I[sel].cpu().numpy(), oi), what
        np.testing.assert_allclose(K64[sel].cpu().numpy(), od, rtol=1e-9, atol=1e-9, err_msg=str(what))
        del idx, rows, stored
