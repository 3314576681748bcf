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
