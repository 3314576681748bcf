"""CPU check of k_logmel_fft_clip's design (csrc/logmel_fft.inc) without a GPU: a numpy restatement of the kernel's data flow --
8 lanes per frame, 25-point FFTs per lane, per-lane twiddles, three exchange stages `own <- (own + g partner)(c + i s)`, the
real-input split against the mirrored lane, the sparse mel accumulation -- driven by the lane tables the LIBRARY builds
(radad_embed_fft_tables: host arithmetic of the shipped .so) and compared with numpy.fft.rfft and the oracle's log-mel
(feature_extraction_whisper.py:135-168).  What this pins: the tables, the lane / register maps and the chunk arithmetic; the
HIP arithmetic itself is checked on the GPU by tests/test_gpu_embed.py and tests/test_gpu_shared_frames.py."""
import ctypes as C

import numpy as np
import pytest

from oracle import radad_oracle as O

TABK, NET, COMB, NTAB = 80, 2000, 2064, 2224


@pytest.fixture(scope="module")
def lib():
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    return _lib.load()


@pytest.fixture(scope="module")
def tables(lib):
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.feature_extractor import mel_filter_bank_slaney
    fb = np.ascontiguousarray(mel_filter_bank_slaney(), np.float32)
    tab = np.zeros(NTAB, np.float32)
    info = (C.c_int32 * 4)()
    assert lib.radad_embed_fft_tables(fb.ctypes.data, tab.ctypes.data, tab.size, info) == 0
    assert info[0] == 1 and info[1] == NTAB and info[3] == COMB
    return fb, tab, list(info)


def _pos(k1):
    return 5 * (k1 % 5) + k1 // 5


def _fft25_inplace(x):
    """the kernel's two in-place radix-5 passes over index m = 5 m1 + m2; Y[k1] ends at _pos(k1)"""
    W5 = np.exp(-2j * np.pi / 5)
    x = x.copy()
    for m2 in range(5):
        v = x[m2::5].copy()
        x[m2::5] = [sum(v[m1] * W5 ** (m1 * j1) for m1 in range(5)) for j1 in range(5)]
    for j1 in range(1, 5):
        for m2 in range(1, 5):
            x[5 * j1 + m2] *= np.exp(-2j * np.pi * (m2 * j1) / 25)
    for j1 in range(5):
        v = x[5 * j1:5 * j1 + 5].copy()
        x[5 * j1:5 * j1 + 5] = [sum(v[m2] * W5 ** (m2 * j2) for m2 in range(5)) for j2 in range(5)]
    return x


def emulate_frame(y, tab, info):
    """y: 400 samples of one frame (pivot already subtracted) -> (linear mel'[80], Re X'[1], X[0..199]) as the 8 lanes compute them"""
    tab64 = tab.astype(np.float64)
    ibits = tab.view(np.int32)
    x = np.zeros((8, 25), complex)
    for p in range(8):
        rr = p if p < 4 else 11 - p
        for m in range(25):
            w0, w1 = tab64[m * TABK + 2 * p], tab64[m * TABK + 2 * p + 1]
            x[p, m] = y[16 * m + 2 * rr] * w0 + 1j * y[16 * m + 2 * rr + 1] * w1
        x[p] = _fft25_inplace(x[p])
        for k1 in range(1, 25):
            c, s = tab64[k1 * TABK + 16 + 2 * p], tab64[k1 * TABK + 16 + 2 * p + 1]
            x[p, _pos(k1)] *= (c - 1j * s)
    partner = (lambda p: 7 - p, lambda p: p ^ 2, lambda p: p ^ 1)
    for stage in range(3):
        new = np.zeros_like(x)
        for p in range(8):
            net = tab64[NET + 8 * p:NET + 8 * p + 8]
            g = net[6] if stage == 2 else net[3 * stage]
            tw = 1.0 if stage == 2 else net[3 * stage + 1] + 1j * net[3 * stage + 2]
            new[p] = (x[p] + g * x[partner[stage](p)]) * tw
        x = new
    row = np.zeros(112)                                       # the frame's row of the mel tile (zeroed by the kernel)
    X = np.zeros(200, complex)
    x1re = None
    for p in range(8):
        k2 = ((p & 1) << 2) | (p & 2) | (p >> 2)
        perm0 = p if p < 2 else (p ^ 1 if p < 4 else 11 - p)
        a0 = a1 = 0.0
        soff = 0
        for k1 in range(25):
            a = x[p, _pos(k1)]
            b = x[perm0, _pos(0)] if k1 == 0 else x[7 - p, _pos(25 - k1)]
            cw, sw = tab64[k1 * TABK + 32 + 2 * p], tab64[k1 * TABK + 32 + 2 * p + 1]
            e1, o1, e2, o2 = a.real + b.real, a.real - b.real, a.imag + b.imag, a.imag - b.imag
            re2 = e1 + cw * e2 - sw * o1
            im2 = o2 - cw * o1 - sw * e2
            pw4 = re2 * re2 + im2 * im2
            wl, wh = tab64[k1 * TABK + 48 + 4 * p], tab64[k1 * TABK + 48 + 4 * p + 1]
            soff = int(ibits[k1 * TABK + 48 + 4 * p + 2])
            adv = int(ibits[k1 * TABK + 48 + 4 * p + 3])
            a0 += pw4 * wl
            a1 += pw4 * wh
            assert soff % 4 == 0 and soff // 4 < 110
            row[soff // 4] = a0                               # the last store of a slot is the complete sum
            if (info[2] >> k1) & 1:                           # some lane's band index rises by two after this bin: both sums are stored
                row[soff // 4 + 1] = a1
            else:
                assert adv < 2
            a0, a1 = (a0, a1) if adv == 0 else ((a1, 0.0) if adv == 1 else (0.0, 0.0))
            X[25 * k2 + k1] = 0.5 * (re2 + 1j * im2)
            if p == 0 and k1 == 1:
                x1re = 0.5 * re2
        row[soff // 4 + 1] = a0
    assert row[111] == 0.0
    comb = ibits[COMB:COMB + 160].reshape(80, 2) // 4
    mel = row[comb[:, 0]] + row[comb[:, 1]]
    return mel, x1re, X


def test_lane_tables_reproduce_rfft_and_the_sparse_mel(tables):
    fb, tab, info = tables
    rng = np.random.default_rng(7)
    hann = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 400)
    for trial in range(3):
        y = rng.standard_normal(400) * (10.0 ** rng.uniform(-3, 3))
        if trial == 2:
            y += 3.0 * np.cos(2 * np.pi * 37.3 * np.arange(400) / 400)          # a strong tone: the bank's dynamic range
        mel, x1re, X = emulate_frame(y, tab, info)
        ref = np.fft.rfft(hann * y)
        scale = np.abs(ref).max()
        assert np.abs(X - ref[:200]).max() < 2e-6 * scale                  # (tables are float32)
        assert abs(x1re - ref[1].real) < 2e-6 * scale
        mel_ref = (np.abs(ref) ** 2) @ fb.astype(np.float64)
        assert np.abs(mel - mel_ref).max() < 1e-5 * mel_ref.max()
        assert np.abs(np.log10(np.maximum(mel, 1e-300)) - np.log10(np.maximum(mel_ref, 1e-300))).max() < 1e-4


def test_bin_200_and_bin_0_carry_no_weight_in_the_slaney_bank(tables):
    fb, tab, info = tables
    assert not fb[200].any() and not fb[0].any()     # what the kernel skips (bin 200) and what the pivot trick needs (bin 0)


def test_emulated_frames_match_the_oracle_log_mel(tables):
    """whole path on one segment: reflect padding, frames, the emulated lanes, log10 -> oracle's log_mel before its max - 8 clamp"""
    fb, tab, info = tables
    from oracle import synth
    seg = synth.audio(0, 1, 32000, 99)[0].astype(np.float64)
    ref = O.log_mel(seg.astype(np.float32))                                   # [200, 80] after clamp and (x + 4) / 4
    padded = np.pad(seg, 200, mode="reflect")
    rows = []
    for f in (0, 1, 2, 57, 198, 199):
        mel, _, _ = emulate_frame(padded[160 * f:160 * f + 400], tab, info)
        rows.append(np.log10(np.maximum(mel, 1e-10)))
    rows = np.array(rows)
    got = (np.maximum(rows, (ref * 4 - 4).max() - 8.0) + 4.0) / 4.0
    assert np.abs(got - ref[[0, 1, 2, 57, 198, 199]]).max() < 1e-5


def test_a_non_triangular_bank_is_refused(lib):
    fb = np.zeros((201, 80), np.float32)
    fb[10, 3] = fb[10, 5] = 1.0                                                # a bin feeding two bands that are not adjacent
    tab = np.zeros(NTAB, np.float32)
    info = (C.c_int32 * 4)()
    assert lib.radad_embed_fft_tables(fb.ctypes.data, tab.ctypes.data, tab.size, info) == 0
    assert info[0] == 0
    fb[:] = 0
    fb[10, 7] = fb[11, 6] = 1.0                                                # the band index falls from one bin to the next
    assert lib.radad_embed_fft_tables(fb.ctypes.data, tab.ctypes.data, tab.size, info) == 0 and info[0] == 0
    fb[:] = 0
    fb[200, 79] = 1.0                                                          # weight on bin 200, which the kernel does not compute
    assert lib.radad_embed_fft_tables(fb.ctypes.data, tab.ctypes.data, tab.size, info) == 0 and info[0] == 0


@pytest.mark.parametrize("S,T,H", [(1, 200, 100), (3, 200, 100), (7, 200, 50), (19, 200, 100), (4, 50, 40), (1, 8, 1), (40, 100, 25)])
def test_fft_chunking_covers_every_frame_once(lib, S, T, H):
    out = (C.c_int32 * 5)()
    assert lib.radad_embed_fft_clip_chunks(S, T, H, out) == 0
    n_full, r, e_tail, n_edge_chunks, total = list(out)
    ni, E = (S - 1) * H + T - 3, 3 * S
    assert n_full * 64 + r == ni and 0 <= r < 64
    assert r + e_tail <= 64 and (160 * (r - 1) + 400 if r else 0) + 400 * e_tail <= 160 * 63 + 400      # slots and sample storage
    assert e_tail <= E and n_edge_chunks == -(-(E - e_tail) // 26) and total == n_full + 1 + n_edge_chunks
