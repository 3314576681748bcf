import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def knn_oracle_lib():
    """ctypes handle on oracle/_build/libknn_oracle.so (built on demand with gcc)."""
    import ctypes as C
    import __graft_entry__ as g
    lib = C.CDLL(g.build_oracle())
    lib.knn_oracle_f64.restype = C.c_int
    lib.knn_oracle_f64.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64,
                                   C.c_void_p, C.c_void_p]
    return lib


def c_knn(lib, db, q, k, metric, id_base=0):
    """float64 brute force through the C oracle.  metric: 'L2' | 'IP' (normalise beforehand for cosine)."""
    import numpy as np
    db = np.ascontiguousarray(db, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    D = np.empty((len(q), k), np.float64)
    I = np.empty((len(q), k), np.int64)
    rc = lib.knn_oracle_f64(db.ctypes.data, len(db), q.ctypes.data, len(q), db.shape[1], k, 0 if metric == "L2" else 1,
                            id_base, D.ctypes.data, I.ctypes.data)
    assert rc == 0
    return D, I


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")
