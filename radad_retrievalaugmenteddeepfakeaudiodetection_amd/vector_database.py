"""VectorDatabase -- same call surface as the reference's vector_database.py:8-272, with the faiss index
object replaced by `HipFlatIndex`, a thin handle on the HBM-resident store in libradad_hip.so.

What callers of the reference rely on and still find here:
  * VectorDatabase(config): .add_vectors / .add_vectors_batch / .search_batch / .search / .save / .load,
    attributes .vector_paths / .vector_labels / .vector_metadata, and the raw `.index`
    (pipeline.py:445-446,480,495-506; app.py:67-76).
  * .index.ntotal / .d / .is_trained / .add(np) / .search(np,k) / .reconstruct(i)  (faiss.IndexFlat surface
    used at vector_database.py:124,138,151,169,181 and pipeline.py:465,503).
Index types: 'L2' -> squared-L2 ascending; 'IP' -> inner product descending, with rows and queries
L2-normalised first when config.normalize_for_ip (default True) i.e. cosine (vector_database.py:61-64,97,100-105).
'IVF' (vector_database.py:65-70) -> HipIVFFlatIndex (k-means coarse quantiser + list scan, csrc/ivf.inc; k <= 128: above 26 the exact scan
of the same rows answers).
Device-resident variants (`search_device`, `reconstruct_batch`) let the pipeline skip the D2H/H2D hops.
"""
import ctypes as C
import hashlib
import logging
import os
import pickle
from typing import Dict, List, Tuple

import numpy as np

from . import _lib


def path_tag(path: str) -> int:
    """Stable 63-bit tag of a file's basename: the device-side stand-in for the strings the reference compares in
    retrieve_similar_vectors (pipeline.py:463,495-502)."""
    h = hashlib.blake2b(os.path.basename(path).encode("utf-8", "surrogatepass"), digest_size=8).digest()
    return int.from_bytes(h, "little") & 0x7FFFFFFFFFFFFFFF


# ---- faiss IndexFlat file interchange -------------------------------------------------------------------------------
# The reference persists its store with faiss.write_index (vector_database.py:203) -> `faiss_index.bin`.  faiss is not
# available here, so this reader/writer follows the published on-disk layout of a flat index from upstream faiss
# (impl/index_write.cpp: fourcc 'IxF2' (L2) / 'IxFI' (IP), then the index header d:int32, ntotal:int64, two int64
# dummies, is_trained:uint8, metric_type:int32 [+ metric_arg:float32 if metric_type > 1], then the row data as a
# uint64 count of 4-byte words followed by ntotal*d float32).  UNVERIFIED against a real faiss build (none in this
# image): round-trips through this module only; treat as best effort for migrating an existing RADAD deployment.
def read_faiss_flat(path: str):
    """-> (metric 'L2'|'IP', d, rows float32 [ntotal, d]); raises ValueError if the file is not a flat faiss index."""
    with open(path, "rb") as f:
        fourcc = f.read(4)
        if fourcc not in (b"IxF2", b"IxFI"):
            raise ValueError(f"{path}: not a faiss IndexFlat file (fourcc {fourcc!r})")
        d = int(np.frombuffer(f.read(4), "<i4")[0])
        ntotal = int(np.frombuffer(f.read(8), "<i8")[0])
        f.read(16)                                   # two int64 dummies
        f.read(1)                                    # is_trained
        metric_type = int(np.frombuffer(f.read(4), "<i4")[0])
        if metric_type > 1:
            f.read(4)                                # metric_arg
        nwords = int(np.frombuffer(f.read(8), "<u8")[0])
        if d <= 0 or ntotal < 0 or nwords != ntotal * d:
            raise ValueError(f"{path}: inconsistent faiss header (d={d}, ntotal={ntotal}, words={nwords})")
        rows = np.fromfile(f, "<f4", nwords).reshape(ntotal, d)
    return ("L2" if fourcc == b"IxF2" else "IP"), d, rows


def write_faiss_flat(path: str, rows: np.ndarray, metric: str):
    rows = np.ascontiguousarray(rows, "<f4")
    n, d = rows.shape
    with open(path, "wb") as f:
        f.write(b"IxF2" if metric.upper() == "L2" else b"IxFI")
        f.write(np.int32(d).tobytes()); f.write(np.int64(n).tobytes())
        f.write(np.int64(1 << 20).tobytes() * 2)
        f.write(b"\x01")
        f.write(np.int32(1 if metric.upper() == "L2" else 0).tobytes())     # faiss MetricType: 0 = IP, 1 = L2
        f.write(np.uint64(n * d).tobytes())
        rows.tofile(f)


def knn_options_from_config(config):
    """the optional scan knobs, read the way the reference reads its own optional ones (getattr with a default,
    vector_database.py:43,67,80): knn_hi_plane, knn_centre, knn_smallq_hi, knn_wide_min_q, knn_dense, knn_live_floor; absent / None = the library's default"""
    opts = dict(hi_plane=getattr(config, "knn_hi_plane", None), centre=getattr(config, "knn_centre", None),
                smallq_hi=getattr(config, "knn_smallq_hi", None), wide_min_q=getattr(config, "knn_wide_min_q", None),
                dense=getattr(config, "knn_dense", None), live_floor=getattr(config, "knn_live_floor", None))
    return {k: v for k, v in opts.items() if v is not None}


class HipFlatIndex:
    """Flat (exhaustive) index living in HBM.  Mirrors the slice of faiss.IndexFlat{L2,IP} the reference uses."""

    is_trained = True   # flat indexes need no training (vector_database.py:124)

    def __init__(self, d: int, metric: int, device: int = 0, id_base: int = 0, store_f16: bool = False, hi_plane=None, centre=None,
                 smallq_hi=None, wide_min_q=None, dense=None, live_floor=None):
        """hi_plane / centre / smallq_hi / wide_min_q / dense / live_floor: kernel choices of the handle (radad_knn_set_option; None = the library's
        default).  They change speed, never results: A/B measurements and the parity tests select kernels through them."""
        self._lib = _lib.load()
        self.d = int(d)
        self.metric = int(metric)
        self.device = int(device)
        self.id_base = int(id_base)
        self.store_f16 = bool(store_f16)      # config.use_float16 (vector_database.py:80): rows kept as IEEE fp16
        h = C.c_void_p()
        _lib.check(self._lib.radad_knn_create_ex(self.d, self.metric, _lib.STORE_F16 if self.store_f16 else _lib.STORE_F32,
                                                 self.device, self.id_base, C.byref(h)), "radad_knn_create")
        self._h = h
        self.options = dict(hi_plane=hi_plane, centre=centre, smallq_hi=smallq_hi, wide_min_q=wide_min_q, dense=dense, live_floor=live_floor)
        for opt, val in ((_lib.KNN_OPT_HI_PLANE, hi_plane), (_lib.KNN_OPT_CENTRE, centre), (_lib.KNN_OPT_SMALLQ_HI, smallq_hi),
                         (_lib.KNN_OPT_WIDE_MIN_Q, wide_min_q), (_lib.KNN_OPT_DENSE, dense), (_lib.KNN_OPT_LIVE_FLOOR, live_floor)):
            if val is not None:
                _lib.check(self._lib.radad_knn_set_option(self._h, opt, int(val)), "radad_knn_set_option")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.radad_knn_destroy(h)
            except Exception:
                pass

    @property
    def ntotal(self) -> int:
        n = C.c_int64()
        _lib.check(self._lib.radad_knn_ntotal(self._h, C.byref(n)))
        return n.value

    def reserve(self, capacity: int):
        _lib.check(self._lib.radad_knn_reserve(self._h, int(capacity)), "radad_knn_reserve")

    # ---- host (numpy) surface: what faiss offers ---------------------------------------------------------
    def add(self, x: np.ndarray):
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add expects [n, {self.d}] float32, got {x.shape}")
        _lib.check(self._lib.radad_knn_add_host(self._h, x.ctypes.data, x.shape[0]), "radad_knn_add_host")

    def search(self, x: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"search expects [nq, {self.d}] float32, got {x.shape}")
        D = np.empty((x.shape[0], k), np.float32)
        I = np.empty((x.shape[0], k), np.int64)
        _lib.check(self._lib.radad_knn_search_host(self._h, x.ctypes.data, x.shape[0], int(k), D.ctypes.data, I.ctypes.data),
                   "radad_knn_search_host")
        return D, I

    def reconstruct(self, i: int) -> np.ndarray:
        idx = np.asarray([int(i)], np.int64)
        out = np.empty((1, self.d), np.float32)
        _lib.check(self._lib.radad_knn_reconstruct_host(self._h, idx.ctypes.data, 1, out.ctypes.data))
        return out[0]

    # ---- device (torch) surface ---------------------------------------------------------------------------
    def add_device(self, x):
        import torch
        _lib.require_cuda(x, "x")
        x = x.contiguous().float()
        if x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"add expects [n, {self.d}], got {tuple(x.shape)}")
        with torch.cuda.device(x.device):
            _lib.check(self._lib.radad_knn_add(self._h, x.data_ptr(), x.shape[0], _lib.stream_ptr(x.device)), "radad_knn_add")

    def search_device(self, q, k: int, return_f64: bool = False):
        """q [nq, d] CUDA tensor (float32, or bfloat16 = BASELINE config 5's bf16 embeddings, handed to the library as
        they are) -> (D f32 [nq,k], I i64 [nq,k]) on the device; with return_f64 also the float64 distances the ranking
        was made on (used by the sharded merge)."""
        import torch
        _lib.require_cuda(q, "q")
        bf16 = q.dtype == torch.bfloat16
        q = q.contiguous() if bf16 else q.contiguous().float()
        if q.dim() != 2 or q.shape[1] != self.d:
            raise ValueError(f"search expects [nq, {self.d}], got {tuple(q.shape)}")
        D = torch.empty((q.shape[0], k), device=q.device, dtype=torch.float32)
        I = torch.empty((q.shape[0], k), device=q.device, dtype=torch.int64)
        K64 = torch.empty((q.shape[0], k), device=q.device, dtype=torch.float64) if return_f64 else None
        with torch.cuda.device(q.device):
            _lib.check(self._lib.radad_knn_search_ex(self._h, q.data_ptr(), _lib.Q_BF16 if bf16 else _lib.Q_F32, q.shape[0], int(k),
                                                     D.data_ptr(), I.data_ptr(), K64.data_ptr() if return_f64 else None,
                                                     _lib.stream_ptr(q.device)), "radad_knn_search")
        return (D, I, K64) if return_f64 else (D, I)

    def search_begin(self, q, k: int):
        """first half of a search over a row shard (sharded.py): prepares the queries and scans this shard; returns a float32 CUDA
        tensor [nq, k] -- per query lower bounds of the exact scores of this shard's k best rows (q.y / -|q - y|^2; -inf where
        there is none).  The caller gathers them from all shards; the k-th largest of a query's G k values (global_bound below) is
        what search_finish takes."""
        import torch
        _lib.require_cuda(q, "q")
        bf16 = q.dtype == torch.bfloat16
        q = q.contiguous() if bf16 else q.contiguous().float()
        if q.dim() != 2 or q.shape[1] != self.d:
            raise ValueError(f"search expects [nq, {self.d}], got {tuple(q.shape)}")
        lb = torch.empty((q.shape[0], int(k)), device=q.device, dtype=torch.float32)
        with torch.cuda.device(q.device):
            _lib.check(self._lib.radad_knn_search_begin(self._h, q.data_ptr(), _lib.Q_BF16 if bf16 else _lib.Q_F32, q.shape[0], int(k),
                                                        lb.data_ptr(), _lib.stream_ptr(q.device)), "radad_knn_search_begin")
        self._begun = (q, int(k))          # (keeps the queries alive until the second half has read them)
        return lb

    @staticmethod
    def global_bound(lb_all, k: int):
        """lb_all [G, nq, k] (every shard's search_begin result) -> [nq]: the k-th largest of each query's G k lower bounds = a lower
        bound of the exact k-th best score over all shards"""
        import torch
        G, nq, kk = lb_all.shape
        if G * kk > HipFlatIndex.KTH_MAX_VALUES or not lb_all.is_cuda:     # beyond the selection kernel's reach (8 ranks x k > 160): torch
            return torch.topk(lb_all.permute(1, 0, 2).reshape(nq, -1).float(), int(k), dim=1).values[:, int(k) - 1].contiguous()
        flat = lb_all.contiguous().float()                 # [G][nq][k] as gathered: the kernel reads that layout
        out = torch.empty((nq,), device=flat.device, dtype=torch.float32)
        with torch.cuda.device(flat.device):
            _lib.check(_lib.load().radad_kth_largest(flat.data_ptr(), nq, G, kk, int(k), out.data_ptr(), flat.device.index,
                                                     _lib.stream_ptr(flat.device)), "radad_kth_largest")
        return out

    def search_finish(self, global_lb=None, return_f64: bool = False):
        """second half: float64 re-rank of what can still be among the GLOBAL k best (global_lb: the element-wise maximum of the
        shards' search_begin bounds; None = this shard's own top k), then the exact kernel for uncertified queries.
        -> (D, I[, K64]) as search_device; with a bound a row may hold fewer than k real entries (-1 filled)."""
        import torch
        if getattr(self, "_begun", None) is None:
            raise ValueError("search_finish: no search was begun on this index (search_begin first; a begun search is finished once)")
        q, k = self._begun
        self._begun = None
        D = torch.empty((q.shape[0], k), device=q.device, dtype=torch.float32)
        I = torch.empty((q.shape[0], k), device=q.device, dtype=torch.int64)
        K64 = torch.empty((q.shape[0], k), device=q.device, dtype=torch.float64) if return_f64 else None
        if global_lb is not None:
            _lib.require_cuda(global_lb, "global_lb")
            global_lb = global_lb.contiguous().float()
            if global_lb.numel() != q.shape[0]:
                raise ValueError("global_lb must hold one bound per query")
        with torch.cuda.device(q.device):
            _lib.check(self._lib.radad_knn_search_finish(self._h, global_lb.data_ptr() if global_lb is not None else None, D.data_ptr(),
                                                         I.data_ptr(), K64.data_ptr() if return_f64 else None,
                                                         _lib.stream_ptr(q.device)), "radad_knn_search_finish")
        return (D, I, K64) if return_f64 else (D, I)

    def search_abort(self):
        """gives up a begun search (the exchange between the shards failed): the index accepts searches again"""
        self._begun = None
        _lib.check(self._lib.radad_knn_search_abort(self._h), "radad_knn_search_abort")

    KTH_MAX_VALUES = 1280        # radad_kth_largest: groups x per_group values per query (csrc/knn.hip KTH_PER_LANE x 64)

    def reconstruct_batch(self, idx):
        """idx: int64 CUDA tensor of any shape -> [*idx.shape, d]; negative ids give zero rows."""
        import torch
        _lib.require_cuda(idx, "idx")
        flat = idx.contiguous().to(torch.int64).reshape(-1)
        out = torch.empty((flat.numel(), self.d), device=idx.device, dtype=torch.float32)
        with torch.cuda.device(idx.device):
            _lib.check(self._lib.radad_knn_reconstruct(self._h, flat.data_ptr(), flat.numel(), out.data_ptr(),
                                                       _lib.stream_ptr(idx.device)), "radad_knn_reconstruct")
        return out.reshape(*idx.shape, self.d)

    def plane_info(self):
        """{"built", "centred", "one_scale", "rebuilds"} of the f16 plane the certified scans read (radad_knn_plane_info, _plane_rebuilds)"""
        b, c, o, r = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.radad_knn_plane_info(self._h, C.byref(b), C.byref(c), C.byref(o)))
        _lib.check(self._lib.radad_knn_plane_rebuilds(self._h, C.byref(r)))
        return {"built": bool(b.value), "centred": bool(c.value), "one_scale": bool(o.value), "rebuilds": r.value}

    def last_emitted(self, nq: int):
        """(candidates emitted per query int32 [nq], final admission floors float32 [nq]) of the last tile-scan search (radad_knn_last_emitted)"""
        c = np.empty(nq, np.int32)
        f = np.empty(nq, np.float32)
        _lib.check(self._lib.radad_knn_last_emitted(self._h, c.ctypes.data, f.ctypes.data, int(nq)), "radad_knn_last_emitted")
        return c, f

    def tuning_info(self):
        """{"cap_boost", "fp32_searches_left", "reports_consumed"} (radad_knn_tuning_info): read without synchronising"""
        a, b, n = C.c_int(), C.c_int(), C.c_int64()
        _lib.check(self._lib.radad_knn_tuning_info(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return {"cap_boost": a.value, "fp32_searches_left": b.value, "reports_consumed": n.value}

    def last_launch(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.radad_knn_last_launch(self._h, C.byref(a), C.byref(b), C.byref(c)))
        st, nq = (C.c_int * 6)(), C.c_int64()
        _lib.check(self._lib.radad_knn_last_certificate(self._h, C.byref(nq), st))
        kind, nl, nph = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.radad_knn_last_scan_kind(self._h, C.byref(kind)))
        _lib.check(self._lib.radad_knn_last_scan_launches(self._h, C.byref(nl)))
        _lib.check(self._lib.radad_knn_last_scan_phases(self._h, C.byref(nph)))
        return {"query_tiles": a.value, "db_splits": b.value, "block_threads": c.value, "rechecked_queries": st[0],
                "scan_launches": nl.value, "scan_phases": nph.value,
                "scan_kind": ("f32_tile", "hi_tile", "f32_smallq", "hi_smallq", "f16_tile", "f32_dense")[kind.value],
                "certificate": {"queries": nq.value, "rejected": st[0], "candidates_rescored": st[1],
                                "rejected_buffer_full": st[2], "rejected_list_used_up": st[3],
                                "rejected_floor_above_threshold": st[4], "rejected_scan_dropped": st[5]}}

    def profile(self, enable: bool = True, every: int = 1):
        """record HIP events around every scan-kernel launch (ring of 64) of every `every`-th search"""
        _lib.check(self._lib.radad_knn_profile(self._h, max(1, int(every)) if enable else 0), "radad_knn_profile")

    def profile_read(self):
        """scan-kernel durations in ms (synchronises on the recorded events)"""
        buf = (C.c_float * 64)()
        n = C.c_int()
        _lib.check(self._lib.radad_knn_profile_read(self._h, buf, 64, C.byref(n)), "radad_knn_profile_read")
        return [float(buf[i]) for i in range(n.value)]

    def save(self, path: str):
        _lib.check(self._lib.radad_knn_save(self._h, os.fsencode(path)), "radad_knn_save")

    def load(self, path: str, row0: int = 0, n_rows: int = -1):
        """replace the contents with rows [row0, row0+n_rows) of a snapshot (default: all of it).  A rank of a sharded
        run loads its own slice of one shared file: HipFlatIndex.load_shard."""
        _lib.check(self._lib.radad_knn_load_range(self._h, os.fsencode(path), int(row0), int(n_rows)), "radad_knn_load")

    @staticmethod
    def snapshot_info(path: str) -> dict:
        """header of a snapshot file: {"d", "metric", "store_f16", "ntotal"}"""
        d, m, t, n = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        _lib.check(_lib.load().radad_knn_snapshot_info(os.fsencode(path), C.byref(d), C.byref(m), C.byref(t), C.byref(n)),
                   "radad_knn_snapshot_info")
        return {"d": d.value, "metric": m.value, "store_f16": t.value == _lib.STORE_F16, "ntotal": n.value}

    @classmethod
    def load_shard(cls, path: str, rank: int, world_size: int, device: int = 0):
        """this rank's row shard of a snapshot written by any number of GPUs: rows shard_bounds(ntotal, world, rank),
        ids reported globally (id_base = first row).  Only the shard's byte range of the file is read."""
        from .sharded import shard_bounds
        info = cls.snapshot_info(path)
        lo, hi = shard_bounds(info["ntotal"], world_size, rank)
        index = cls(info["d"], info["metric"], device=device, id_base=lo, store_f16=info["store_f16"])
        index.load(path, lo, hi - lo)
        return index


class HipIVFFlatIndex:
    """faiss.IndexIVFFlat(IndexFlatL2(d), d, nlist, METRIC_L2) on the GPU: the slice of its surface the reference uses
    (vector_database.py:65-70,124-128,138,174-181; pipeline.py:503): d, nlist, nprobe, is_trained, ntotal, train, add,
    search, reconstruct."""

    def __init__(self, d: int, nlist: int, device: int = 0, niter: int = 10, hi_scan=None):
        """hi_scan: 0 keeps the list scans on the fp32 rows (radad_ivf_set_option; None = the library's default, the certified f16 scan)"""
        self._lib = _lib.load()
        self.d, self.nlist, self.device, self.niter = int(d), int(nlist), int(device), int(niter)
        self.nprobe = 1                      # faiss default; the reference sets it from config.vector_db_nprobe (:177)
        self.id_base = 0
        self.metric = _lib.METRIC_L2
        self.store_f16 = False
        h = C.c_void_p()
        _lib.check(self._lib.radad_ivf_create(self.d, self.nlist, self.device, C.byref(h)), "radad_ivf_create")
        self._h = h
        if hi_scan is not None:
            _lib.check(self._lib.radad_ivf_set_option(self._h, _lib.IVF_OPT_HI_SCAN, int(hi_scan)), "radad_ivf_set_option")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.radad_ivf_destroy(h)
            except Exception:
                pass

    def _dev(self):
        import torch
        return torch.device("cuda", self.device)

    @property
    def is_trained(self) -> bool:
        t = C.c_int()
        _lib.check(self._lib.radad_ivf_is_trained(self._h, C.byref(t)))
        return bool(t.value)

    @property
    def ntotal(self) -> int:
        n = C.c_int64()
        _lib.check(self._lib.radad_ivf_ntotal(self._h, C.byref(n)))
        return n.value

    def _to_dev(self, x):
        import torch
        if isinstance(x, torch.Tensor):
            return _lib.require_cuda(x, "x").contiguous().float()
        return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(self._dev())

    def train(self, x):
        import torch
        t = self._to_dev(x)
        with torch.cuda.device(t.device):
            _lib.check(self._lib.radad_ivf_train(self._h, t.data_ptr(), t.shape[0], self.niter, _lib.stream_ptr(t.device)), "radad_ivf_train")

    def set_centroids(self, c):
        import torch
        t = self._to_dev(c)
        if tuple(t.shape) != (self.nlist, self.d):
            raise ValueError(f"centroids must be [{self.nlist}, {self.d}]")
        with torch.cuda.device(t.device):
            _lib.check(self._lib.radad_ivf_set_centroids(self._h, t.data_ptr(), _lib.stream_ptr(t.device)), "radad_ivf_set_centroids")

    def centroids(self) -> np.ndarray:
        import torch
        out = torch.empty((self.nlist, self.d), device=self._dev())
        with torch.cuda.device(out.device):
            _lib.check(self._lib.radad_ivf_centroids(self._h, out.data_ptr(), _lib.stream_ptr(out.device)))
        return out.cpu().numpy()

    def assignments(self) -> np.ndarray:
        out = np.empty(self.ntotal, np.int32)
        _lib.check(self._lib.radad_ivf_assignments_host(self._h, out.ctypes.data, out.size))
        return out

    def add(self, x):
        import torch
        t = self._to_dev(x)
        if t.dim() != 2 or t.shape[1] != self.d:
            raise ValueError(f"add expects [n, {self.d}], got {tuple(t.shape)}")
        with torch.cuda.device(t.device):
            _lib.check(self._lib.radad_ivf_add(self._h, t.data_ptr(), t.shape[0], _lib.stream_ptr(t.device)), "radad_ivf_add")

    add_device = add

    MAX_K = 128     # csrc/ivf.inc: up to k = 26 the probed lists are scanned (k + 6 candidates per (query, list) in 32-entry register
                    # lists); 27..128 is answered by the exact certified scan of the same rows (recall 1.0, the flat scan's cost)

    def search_device(self, q, k: int):
        import torch
        if k > self.MAX_K:      # faiss accepts k up to 2048; refusing loudly beats the silent empty result a swallowed error gives
            raise ValueError(f"HipIVFFlatIndex supports k <= {self.MAX_K} (asked for {k}); use a flat index for larger k")
        q = self._to_dev(q)
        D = torch.empty((q.shape[0], k), device=q.device, dtype=torch.float32)
        I = torch.empty((q.shape[0], k), device=q.device, dtype=torch.int64)
        with torch.cuda.device(q.device):
            _lib.check(self._lib.radad_ivf_search(self._h, q.data_ptr(), q.shape[0], int(k), int(self.nprobe), D.data_ptr(), I.data_ptr(),
                                                  _lib.stream_ptr(q.device)), "radad_ivf_search")
        ex = C.c_int()
        _lib.check(self._lib.radad_ivf_last_search_exact(self._h, C.byref(ex)))
        self.last_search_exact = bool(ex.value)
        if self.last_search_exact and not getattr(self, "_warned_exact", False):
            self._warned_exact = True
            logging.getLogger(__name__).warning(
                "HipIVFFlatIndex: k=%d > 26 is answered by the exact scan of all rows (a superset of the nprobe=%d lists' rows, at the "
                "flat search's cost); faiss.IndexIVFFlat would return rows of the probed lists only", k, self.nprobe)
        return D, I

    def search(self, x, k: int):
        D, I = self.search_device(x, k)
        return D.cpu().numpy(), I.cpu().numpy()

    def last_search_info(self) -> dict:
        """{"scan": "f32_lists" | "hi_lists" | "exact_flat", "rejected": queries the f16 scan's certificate handed to the fp32 pass}
        of the most recent search (synchronises with it)"""
        kind, rej = C.c_int(), C.c_int()
        _lib.check(self._lib.radad_ivf_last_search_info(self._h, C.byref(kind), C.byref(rej)), "radad_ivf_last_search_info")
        return {"scan": _lib.IVF_SCAN_KINDS[kind.value], "rejected": rej.value}

    SNAPSHOT_CHUNK = 1 << 18

    def save(self, path: str):
        """rows in insertion order as a native flat snapshot at `path` (gathered on the device chunk by chunk into a scratch
        flat store and streamed out by radad_knn_save: the store never exists in host memory), centroids at path + '.ivf.npz'"""
        import torch
        flat = HipFlatIndex(self.d, _lib.METRIC_L2, device=self.device)
        n = self.ntotal
        flat.reserve(max(n, 1))
        for s0 in range(0, n, self.SNAPSHOT_CHUNK):
            ids = torch.arange(s0, min(n, s0 + self.SNAPSHOT_CHUNK), device=self._dev())
            flat.add_device(self.reconstruct_batch(ids))
        flat.save(path)
        with open(path + ".ivf.npz", "wb") as f:
            np.savez(f, kind="radad_ivf", centroids=self.centroids(), nlist=self.nlist, ntotal=n)

    def load(self, path: str):
        """inverse of save(): the centroids are installed as they were, the rows are re-added in insertion order (so ids and list
        contents come out as before); the snapshot is streamed to the device by radad_knn_load"""
        import torch
        z = np.load(path + ".ivf.npz")
        centroids = z["centroids"]
        if centroids.shape != (self.nlist, self.d):
            raise ValueError(f"snapshot has {centroids.shape[0]} lists of dimension {centroids.shape[1]}, index has {self.nlist} x {self.d}")
        if self.ntotal:
            raise ValueError("load() needs an empty IVF index")
        self.set_centroids(centroids)
        flat = HipFlatIndex(self.d, _lib.METRIC_L2, device=self.device)
        flat.load(path)
        if flat.ntotal != int(z["ntotal"]):
            raise ValueError(f"{path} holds {flat.ntotal} rows, its sidecar says {int(z['ntotal'])}")
        for s0 in range(0, flat.ntotal, self.SNAPSHOT_CHUNK):
            ids = torch.arange(s0, min(flat.ntotal, s0 + self.SNAPSHOT_CHUNK), device=self._dev())
            self.add(flat.reconstruct_batch(ids))

    def reconstruct_batch(self, idx):
        import torch
        _lib.require_cuda(idx, "idx")
        flat = idx.contiguous().to(torch.int64).reshape(-1)
        out = torch.empty((flat.numel(), self.d), device=idx.device, dtype=torch.float32)
        with torch.cuda.device(idx.device):
            _lib.check(self._lib.radad_ivf_reconstruct(self._h, flat.data_ptr(), flat.numel(), out.data_ptr(), _lib.stream_ptr(idx.device)))
        return out.reshape(*idx.shape, self.d)

    def reconstruct(self, i: int) -> np.ndarray:
        import torch
        return self.reconstruct_batch(torch.tensor([int(i)], device=self._dev()))[0].cpu().numpy()


class VectorDatabase:
    """Store of clip embeddings + brute-force retrieval (vector_database.py:8-272)."""

    def __init__(self, config):
        self.config = config
        self.index = None
        self.vector_paths: List[str] = []
        self.vector_labels: List = []
        self.vector_metadata: Dict[str, list] = {}
        self.db_path = os.path.join(config.vector_db_path, "faiss_index.bin")   # file names kept (vector_database.py:18-19)
        self.metadata_path = os.path.join(config.vector_db_path, "metadata.pkl")
        self.device_id = 0
        os.makedirs(config.vector_db_path, exist_ok=True)
        self._cosine = False
        self._tags_host: List[int] = []      # path_tag of every stored row (device copy built lazily)
        self._tag_names: Dict[int, str] = {}  # tag -> basename, to assert the tags are collision-free
        self._tags_dev = None
        self._labels_dev = None
        import torch
        dev = torch.device(getattr(config, "device", "cuda"))
        if dev.type == "cuda":
            self.device_id = dev.index if dev.index is not None else torch.cuda.current_device()

    # vector_database.py:56-97
    def create_index(self, dimension: int, id_base: int = 0):
        index_type = self.config.vector_db_index_type.upper()
        self._cosine = (index_type == "IP") and bool(getattr(self.config, "normalize_for_ip", True))
        if index_type == "L2":
            metric = _lib.METRIC_L2
        elif index_type == "IP":
            metric = _lib.METRIC_COSINE if self._cosine else _lib.METRIC_IP
        elif index_type == "IVF":
            nlist = max(64, int(getattr(self.config, "ivf_nlist", 4096)))                     # vector_database.py:67-68
            self.index = HipIVFFlatIndex(dimension, nlist, self.device_id)
            logging.info(f"Created HIP IVF-flat index on device {self.device_id} dim={dimension} nlist={nlist}")
            return
        else:
            raise ValueError(f"Unsupported index type: {index_type}")
        self.index = HipFlatIndex(dimension, metric, self.device_id, id_base,
                                  store_f16=bool(getattr(self.config, "use_float16", False)),
                                  **knn_options_from_config(self.config))          # vector_database.py:80
        logging.info(f"Created HIP flat index on device {self.device_id} dim={dimension} type={index_type}")

    # vector_database.py:108-151
    def add_vectors_batch(self, vectors, paths: List[str], labels: List[int], metadata: Dict, batch_size: int = 10000):
        if vectors.shape[0] == 0:
            logging.warning("No vectors to add to database")
            return
        if self.index is None:
            self.create_index(vectors.shape[1])
        is_dev = hasattr(vectors, "is_cuda") and vectors.is_cuda
        if not is_dev:
            vectors = np.ascontiguousarray(np.asarray(vectors).astype(np.float32, copy=False))
        try:     # IVF: train once, on a representative prefix, before adding (vector_database.py:122-130)
            if hasattr(self.index, "is_trained") and not self.index.is_trained:
                logging.info("Training IVF index...")
                self.index.train(vectors[:min(50000, vectors.shape[0])])
        except Exception as e:
            logging.warning(f"Index training skipped/failed: {e}")
        total = vectors.shape[0]
        added = 0
        for start in range(0, total, batch_size):
            end = min(start + batch_size, total)
            # the 63-bit tags stand in for the basename strings the reference compares (pipeline.py:495-502): two distinct
            # basenames must never share one.  Checked before anything is added (a collision is a hard error, not a skipped batch).
            batch_tags = []
            for p_ in paths[start:end]:
                t_, b_ = path_tag(p_), os.path.basename(p_)
                if self._tag_names.setdefault(t_, b_) != b_:
                    raise RuntimeError(f"path_tag collision between {self._tag_names[t_]!r} and {b_!r}")
                batch_tags.append(t_)
            try:
                if is_dev:
                    self.index.add_device(vectors[start:end])
                else:
                    self.index.add(vectors[start:end])     # normalisation for cosine happens in the add kernel
                added += end - start
                self._tags_host.extend(batch_tags)
                self._tags_dev = self._labels_dev = None
                self.vector_paths.extend(paths[start:end])
                self.vector_labels.extend(labels[start:end])
                for key, values in metadata.items():
                    self.vector_metadata.setdefault(key, [])
                    vals = values[start:end] if hasattr(values, "__getitem__") else [values] * (end - start)
                    self.vector_metadata[key].extend(vals)
            except Exception as e:   # log-and-skip, as vector_database.py:147-149
                logging.error(f"Error adding batch {start}-{end}: {e}")
                continue
        logging.info(f"Added {added}/{total} vectors. Index ntotal={self.index.ntotal}")

    # vector_database.py:154-157
    def add_vectors(self, vectors, paths: List[str], labels: List[int], metadata: Dict):
        self.add_vectors_batch(vectors, paths, labels, metadata, getattr(self.config, "vector_add_batch_size", 10000))

    # vector_database.py:159-182
    def search_batch(self, query_vectors, k: int = None):
        """numpy in -> (float32 [B,k], int64 [B,k]) numpy out, as the reference; a CUDA tensor in -> CUDA tensors out."""
        if self.index is None:
            raise ValueError("Vector database is empty. Build the database first.")
        k = int(k if k is not None else getattr(self.config, "top_k", 5))
        is_dev = hasattr(query_vectors, "is_cuda") and query_vectors.is_cuda
        if not is_dev:
            query_vectors = np.asarray(query_vectors)
        if query_vectors.ndim == 1:
            query_vectors = query_vectors.reshape(1, -1)
        k = min(k, self.index.ntotal)
        if k <= 0:
            logging.warning("No vectors available for search")
            if is_dev:
                import torch
                return (torch.zeros((len(query_vectors), 0), dtype=torch.float32, device=query_vectors.device),
                        torch.zeros((len(query_vectors), 0), dtype=torch.int64, device=query_vectors.device))
            return np.zeros((len(query_vectors), 0), dtype=np.float32), np.zeros((len(query_vectors), 0), dtype=np.int64)
        try:     # IVF: tune nprobe (vector_database.py:174-179)
            if hasattr(self.index, "nprobe") and hasattr(self.config, "vector_db_nprobe"):
                self.index.nprobe = int(self.config.vector_db_nprobe)
        except Exception:
            pass
        if is_dev:
            return self.index.search_device(query_vectors, k)
        return self.index.search(query_vectors.astype(np.float32, copy=False), k)

    # ---- device-side columns for retrieve_similar_vectors ------------------------------------------------------
    def row_tags_device(self):
        """int64 [ntotal] CUDA tensor: path_tag(vector_paths[i])"""
        import torch
        if self._tags_dev is None or self._tags_dev.numel() != len(self.vector_paths):
            if len(self._tags_host) != len(self.vector_paths):          # e.g. after load()
                self._tags_host, self._tag_names = [], {}
                for p_ in self.vector_paths:
                    t_, b_ = path_tag(p_), os.path.basename(p_)
                    if self._tag_names.setdefault(t_, b_) != b_:
                        raise RuntimeError(f"path_tag collision between {self._tag_names[t_]!r} and {b_!r}")
                    self._tags_host.append(t_)
            self._tags_dev = torch.tensor(self._tags_host, dtype=torch.int64, device=torch.device("cuda", self.device_id))
        return self._tags_dev

    def labels_device(self):
        """float32 [ntotal] CUDA tensor of vector_labels (the reference stacks them as float32, pipeline.py:523)"""
        import torch
        if self._labels_dev is None or self._labels_dev.numel() != len(self.vector_labels):
            self._labels_dev = torch.tensor([float(l) for l in self.vector_labels], dtype=torch.float32,
                                            device=torch.device("cuda", self.device_id))
        return self._labels_dev

    def filter_hits(self, dists, idxs, k_keep: int, exclude_tags=None):
        """first k_keep hits per row whose tag is not excluded; (-1, NaN) padded.  All on the device."""
        import torch
        lib = _lib.load()
        B, k_in = idxs.shape
        out_d = torch.empty((B, k_keep), device=idxs.device, dtype=torch.float32)
        out_i = torch.empty((B, k_keep), device=idxs.device, dtype=torch.int64)
        n_excl = 0 if exclude_tags is None else int(exclude_tags.numel())
        tags = self.row_tags_device() if n_excl else None
        d, i = dists.contiguous().float(), idxs.contiguous()
        with torch.cuda.device(idxs.device):
            _lib.check(lib.radad_filter_topk(d.data_ptr(), i.data_ptr(), B, k_in, k_keep, tags.data_ptr() if n_excl else None,
                                             self.index.ntotal, self.index.id_base, exclude_tags.data_ptr() if n_excl else None,
                                             n_excl, out_d.data_ptr(), out_i.data_ptr(), idxs.device.index,
                                             _lib.stream_ptr(idxs.device)), "radad_filter_topk")
        return out_d, out_i

    # vector_database.py:185-188
    def search(self, query_vector, k: int = None):
        distances, indices = self.search_batch(query_vector.reshape(1, -1), k)
        return (distances[0] if len(distances) > 0 else np.array([]), indices[0] if len(indices) > 0 else np.array([]))

    # vector_database.py:190-216 (own snapshot format instead of faiss.write_index; same metadata keys)
    def save(self):
        try:
            if self.index is None:
                logging.warning("No index to save.")
                return
            # IVF: rows in insertion order as a NATIVE flat snapshot (streamed from HBM by radad_knn_save, no host copy of the
            # store), centroids in a small sidecar; assignments are recomputed on load (deterministic: nearest centroid)
            self.index.save(self.db_path)
            if not isinstance(self.index, HipIVFFlatIndex) and os.path.exists(self.db_path + ".ivf.npz"):
                os.remove(self.db_path + ".ivf.npz")      # centroids of an IVF store saved here earlier: not this store's
            meta = {"paths": self.vector_paths, "labels": self.vector_labels, "metadata": self.vector_metadata,
                    "index_type": self.config.vector_db_index_type, "dimension": self.index.d}
            with open(self.metadata_path, "wb") as f:
                pickle.dump(meta, f)
            logging.info(f"Saved index to {self.db_path} with {self.index.ntotal} vectors")
        except Exception as e:
            logging.error(f"Error saving vector database: {e}")

    # vector_database.py:218-242
    def load(self, shard=None):
        """shard=(rank, world_size): keep only this rank's row shard of the saved store (native snapshots only) -- the index
        reads just its byte range of the file and reports global ids; paths / labels / metadata are cut to the same rows."""
        try:
            if not (os.path.exists(self.db_path) and os.path.exists(self.metadata_path)):
                logging.warning("No saved vector database found")
                return
            with open(self.metadata_path, "rb") as f:
                meta = pickle.load(f)
            self.vector_paths, self.vector_labels, self.vector_metadata = meta["paths"], meta["labels"], meta["metadata"]
            self._tags_host, self._tags_dev, self._labels_dev = [], None, None      # device columns follow the new rows
            with open(self.db_path, "rb") as f:
                magic = f.read(8)
            if shard is not None and magic != b"RADADKNN":
                raise ValueError("sharded load is only available for native flat snapshots")
            # which kind of store the snapshot holds is what save() recorded (meta["index_type"]), not whether a centroid sidecar
            # happens to lie beside it (files written before the key existed: the sidecar decides, as it used to)
            saved_type = str(meta.get("index_type", "IVF" if os.path.exists(self.db_path + ".ivf.npz") else "")).upper()
            if magic == b"RADADKNN" and saved_type == "IVF":      # an IVF store written by save() above
                if shard is not None:
                    raise ValueError("sharded load is only available for flat stores")
                self.create_index(int(meta["dimension"]))
                if not isinstance(self.index, HipIVFFlatIndex):
                    raise ValueError("saved IVF store does not match config (index type)")
                self.index.load(self.db_path)
            elif magic[:2] == b"PK":                    # numpy .npz: an IVF store written by an earlier version of save()
                z = np.load(self.db_path)
                centroids, rows = z["centroids"], z["rows"]        # NpzFile re-reads an array on every access: read once
                self.create_index(int(centroids.shape[1]))
                if not isinstance(self.index, HipIVFFlatIndex) or self.index.nlist != centroids.shape[0]:
                    raise ValueError("saved IVF store does not match config (index type / ivf_nlist)")
                self.index.set_centroids(centroids)
                for s0 in range(0, len(rows), 1 << 18):
                    self.index.add(rows[s0:s0 + (1 << 18)])
            elif magic[:4] in (b"IxF2", b"IxFI"):
                # a store written by the reference itself (faiss.write_index).  For cosine the reference had normalised
                # the rows before adding them (vector_database.py:118); adding them again re-normalises unit rows, which
                # changes them by at most one ulp.
                file_metric, d, rows = read_faiss_flat(self.db_path)
                want = self.config.vector_db_index_type.upper()
                if want in ("L2", "IP") and want != file_metric:     # faiss.read_index keeps the file's metric; so do we
                    raise ValueError(f"{self.db_path} holds a faiss Index{file_metric} store but config asks for {want}")
                self.create_index(d)
                for s0 in range(0, len(rows), 1 << 18):
                    self.index.add(rows[s0:s0 + (1 << 18)])
            elif shard is not None:
                from .sharded import shard_bounds
                rank, world = int(shard[0]), int(shard[1])
                self.create_index(int(meta["dimension"]))
                if not isinstance(self.index, HipFlatIndex):
                    raise ValueError("sharded load needs a flat store")
                lo, hi = shard_bounds(HipFlatIndex.snapshot_info(self.db_path)["ntotal"], world, rank)
                self.index = HipFlatIndex(self.index.d, self.index.metric, device=self.device_id, id_base=lo,
                                          store_f16=self.index.store_f16, **{k: v for k, v in getattr(self.index, "options", {}).items() if v is not None})
                self.index.load(self.db_path, lo, hi - lo)
                self.vector_paths, self.vector_labels = self.vector_paths[lo:hi], self.vector_labels[lo:hi]
                if isinstance(self.vector_metadata, dict):
                    self.vector_metadata = {key: vals[lo:hi] for key, vals in self.vector_metadata.items()}
            else:
                self.create_index(int(meta["dimension"]))
                self.index.load(self.db_path)
            logging.info(f"Loaded index; ntotal={self.index.ntotal}")
        except Exception as e:
            logging.error(f"Error loading vector database: {e}")

    # vector_database.py:245-256
    def get_gpu_memory_usage(self):
        try:
            import torch
            if torch.cuda.is_available():
                free, total = torch.cuda.mem_get_info(self.device_id)
                used = total - free
                return {"used": int(used), "total": int(total), "utilization": float(used / total)}
        except Exception:
            pass
        return None
