#!/usr/bin/env python3
"""(test infrastructure, run by hand: python tests/fuzz_knn.py [seed] [cases])
Randomised parity sweep of radad_knn_search against the float64 C oracle (oracle/knn_oracle.c): random store sizes,
batch sizes, dims, k, metrics, store dtypes, append patterns and id bases.  Prints one line per case and a verdict."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as G
from oracle import synth
from radad_retrievalaugmenteddeepfakeaudiodetection_amd import HipFlatIndex, _lib

G.build_oracle()
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libknn_oracle.so"))
lib.knn_oracle_f64.restype = C.c_int
lib.knn_oracle_f64.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]

def c_knn(db, q, k, metric, id_base):
    D = np.empty((len(q), k), np.float64); I = np.empty((len(q), k), np.int64)
    assert lib.knn_oracle_f64(db.ctypes.data, len(db), q.ctypes.data, len(q), db.shape[1], k, 0 if metric == "L2" else 1, id_base,
                              D.ctypes.data, I.ctypes.data) == 0
    return D, I

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
bad = 0
for case in range(ncase):
    metric = ["L2", "IP", "COSINE"][rng.integers(3)]
    dim = int(rng.choice([32, 64, 96, 128, 256, 512, 100, 36]))
    n = int(rng.choice([1, 17, 255, 256, 257, 1000, 4097, 20000, 70001, 150000]))
    nq = int(rng.choice([1, 16, 17, 33, 128, 129, 255, 256, 257, 600, 1025]))
    k = int(rng.choice([1, 5, 10, 11, 15, 24, 26, 40]))
    f16 = bool(rng.integers(4) == 0)
    id_base = int(rng.choice([0, 0, 12345678901]))
    db = synth.rows(0, n, dim, 9000 + case)
    q = synth.rows(0, nq, dim, 9500 + case)
    if rng.integers(2) and not f16 and metric != "COSINE":
        db *= np.exp2(rng.integers(-10, 10, size=n)).astype(np.float32)[:, None]
    for j in range(min(nq, 50)):
        db[(j * 31 + 7) % n] = q[j] + np.float32(0.05) * synth.rows(j, 1, dim, 9900 + case)[0]
    m = {"L2": _lib.METRIC_L2, "IP": _lib.METRIC_IP, "COSINE": _lib.METRIC_COSINE}[metric]
    idx = HipFlatIndex(dim, m, 0, id_base, store_f16=f16)
    cut = int(rng.integers(0, n + 1))
    if cut: idx.add(db[:cut])
    if cut and rng.integers(2): idx.search(q[: min(nq, 300)], min(k, cut))       # a search between the appends
    if cut < n: idx.add(db[cut:])
    D, I = idx.search(q, k)
    rec = idx.reconstruct_batch(torch.arange(id_base, id_base + n, device=dev)).cpu().numpy()
    qq = q.astype(np.float64)
    if metric == "COSINE":
        qq = qq / (np.sqrt((qq ** 2).sum(1))[:, None] + 1e-300)
    kk = min(k, n)
    od, oi = c_knn(np.ascontiguousarray(rec, np.float32), np.ascontiguousarray(qq.astype(np.float32) if metric != "COSINE" else qq.astype(np.float32)), kk, "L2" if metric == "L2" else "IP", id_base)
    ok_i = np.array_equal(I[:, :kk], oi)
    if not ok_i and metric == "COSINE":
        # the oracle saw float32-rounded normalised queries; ranking ties at 1e-8 may differ: compare on exact float64 scores
        sc = rec.astype(np.float64) @ qq.T
        ok_i = all(np.allclose(np.sort(sc[I[j, :kk] - id_base, j])[::-1], np.sort(sc[oi[j] - id_base, j])[::-1], rtol=0, atol=1e-7) for j in range(nq))
    pad_ok = bool(np.all(I[:, kk:] == -1))
    info = idx.last_launch()
    line = dict(case=case, metric=metric, dim=dim, n=n, nq=nq, k=k, f16=f16, id_base=id_base, threads=info["block_threads"],
                rechecked=info["rechecked_queries"], ids_ok=bool(ok_i), pad_ok=pad_ok)
    print(json.dumps(line), flush=True)
    bad += 0 if (ok_i and pad_ok) else 1
print("FUZZ", "FAILED" if bad else "ok", bad, "bad of", ncase)
sys.exit(1 if bad else 0)
