"""ctypes binding of libradad_hip.so (the C ABI declared in include/radad_hip.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing or cannot be
loaded this module raises, and every operator that needs it fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RADAD_HIP_LIB: another build of the same library (tools/exp_*.py point it at libradad_hip_exp.so, the build with the
# kernels' timing-ablation switches); the default is the in-tree product library
LIB_PATH = os.environ.get("RADAD_HIP_LIB") or os.path.join(_HERE, "libradad_hip.so")

RADAD_OK, RADAD_EINVAL, RADAD_EHIP, RADAD_ENOMEM, RADAD_EIO, RADAD_ESTATE = 0, -1, -2, -3, -4, -5
METRIC_L2, METRIC_IP, METRIC_COSINE = 0, 1, 2
POOL_MAX, POOL_AVG = 0, 1
STORE_F32, STORE_F16 = 0, 1
Q_F32, Q_BF16 = 0, 1
OUT_F32, OUT_BF16 = 0, 1
MAX_LEVELS = 8
KNN_MAX_K = 1024
KNN_OPT_HI_PLANE, KNN_OPT_CENTRE, KNN_OPT_SMALLQ_HI, KNN_OPT_WIDE_MIN_Q, KNN_OPT_DENSE, KNN_OPT_LIVE_FLOOR = 0, 1, 2, 3, 4, 5
IVF_OPT_HI_SCAN = 0
IVF_SCAN_KINDS = ("f32_lists", "hi_lists", "exact_flat")
EMBED_NO_SHARED_FRAMES, EMBED_LOGMEL_F32, EMBED_LOGMEL_DFT_GEMM = 1, 2, 4

c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)


class EmbedCfg(C.Structure):
    _fields_ = [("segment_length", C.c_int32), ("hop_length", C.c_int32), ("normalize", C.c_int32),
                ("n_fft", C.c_int32), ("fft_hop", C.c_int32), ("n_mels", C.c_int32),
                ("padded_samples", C.c_int32), ("feat_dim", C.c_int32), ("n_levels", C.c_int32),
                ("levels", C.c_int32 * MAX_LEVELS), ("pool_mode", C.c_int32)]


class ProjWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "w5", "b5",
                                          "ln_g", "ln_b", "w6", "b6", "w54t", "b54")]


HEAD_MAX_LAYERS = 6


class HeadWeights(C.Structure):
    _fields_ = [("wf", C.c_void_p), ("bf", C.c_void_p), ("n_layers", C.c_int32),
                ("dims", C.c_int32 * (HEAD_MAX_LAYERS + 1)),
                ("lw", C.c_void_p * HEAD_MAX_LAYERS), ("lb", C.c_void_p * HEAD_MAX_LAYERS),
                ("bn_scale", C.c_void_p * HEAD_MAX_LAYERS), ("bn_shift", C.c_void_p * HEAD_MAX_LAYERS)]


# name -> (restype, argtypes); every symbol include/radad_hip.h declares
SIGNATURES = {
    "radad_abi_version": (C.c_int, []),
    "radad_last_error": (C.c_char_p, []),
    "radad_device_count": (C.c_int, []),
    "radad_knn_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "radad_knn_create_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "radad_knn_destroy": (C.c_int, [C.c_void_p]),
    "radad_knn_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "radad_knn_dim": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_knn_metric": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_knn_ntotal": (C.c_int, [C.c_void_p, c_i64p]),
    "radad_knn_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "radad_knn_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "radad_knn_add_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "radad_knn_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "radad_knn_search_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "radad_knn_search_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "radad_knn_search_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "radad_knn_search_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "radad_knn_search_abort": (C.c_int, [C.c_void_p]),
    "radad_knn_scan_geometry": (C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "radad_knn_last_scan_launches": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_knn_last_scan_phases": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_knn_last_emitted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "radad_knn_plane_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "radad_knn_plane_rebuilds": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_knn_tuning_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "radad_kth_largest": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "radad_knn_last_scan_kind": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_knn_search_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "radad_knn_reconstruct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "radad_knn_reconstruct_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "radad_knn_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "radad_knn_load": (C.c_int, [C.c_void_p, C.c_char_p]),
    "radad_knn_load_range": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64]),
    "radad_knn_snapshot_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int64)]),
    "radad_knn_last_recheck": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_knn_last_certificate": (C.c_int, [C.c_void_p, c_i64p, C.POINTER(C.c_int)]),
    "radad_knn_last_launch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "radad_knn_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "radad_knn_profile_read": (C.c_int, [C.c_void_p, c_f32p, C.c_int, C.POINTER(C.c_int)]),
    "radad_embed_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "radad_embed_profile_read": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.POINTER(C.c_int)]),
    "radad_topk_merge": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p]),
    "radad_topk_merge_f64": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "radad_filter_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "radad_ivf_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "radad_ivf_destroy": (C.c_int, [C.c_void_p]),
    "radad_ivf_is_trained": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_ivf_ntotal": (C.c_int, [C.c_void_p, c_i64p]),
    "radad_ivf_nlist": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_ivf_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "radad_ivf_set_centroids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "radad_ivf_centroids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "radad_ivf_assignments_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "radad_ivf_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "radad_ivf_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "radad_ivf_last_search_exact": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_ivf_last_search_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "radad_ivf_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "radad_ivf_reconstruct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "radad_rownorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "radad_embed_create": (C.c_int, [C.POINTER(EmbedCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.POINTER(C.c_void_p)]),
    "radad_embed_create_ex": (C.c_int, [C.POINTER(EmbedCfg), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_void_p)]),
    "radad_embed_destroy": (C.c_int, [C.c_void_p]),
    "radad_embed_output_dim": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_embed_num_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_segment_count": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "radad_embed_forward": (C.c_int, [C.c_void_p, C.c_void_p, c_i64p, C.c_int64, C.c_void_p, C.c_void_p]),
    "radad_embed_forward_ex": (C.c_int, [C.c_void_p, C.c_void_p, c_i64p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "radad_embed_forward_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "radad_pcm16_to_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "radad_embed_plan_flags": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_embed_plan_flags_poll": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "radad_embed_last_logmel_kind": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "radad_embed_clip_chunks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "radad_embed_fft_clip_chunks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "radad_embed_fft_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "radad_embed_normalize": (C.c_int, [C.c_void_p, C.c_void_p, c_i64p, c_i32p, C.c_int64, C.c_void_p, C.c_void_p]),
    "radad_embed_logmel": (C.c_int, [C.c_void_p, C.c_void_p, c_i64p, c_i32p, C.c_int64, C.c_void_p, C.c_void_p]),
    "radad_embed_frame_features": (C.c_int, [C.c_void_p, C.c_void_p, c_i64p, c_i32p, C.c_int64, C.c_void_p,
                                             C.c_void_p]),
    "radad_tpp_forward": (C.c_int, [C.c_void_p, c_i64p, C.c_int64, C.c_int, c_i32p, C.c_int, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p]),
    "radad_group_mean": (C.c_int, [C.c_void_p, c_i64p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "radad_projection_forward": (C.c_int, [C.POINTER(ProjWeights), C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "radad_projection_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]),
    "radad_projection_fold": (C.c_int, [C.POINTER(ProjWeights), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p]),
    "radad_linear_forward": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int64,
                                       C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int,
                                       C.c_void_p]),
    "radad_linear_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "radad_fuse_head_forward": (C.c_int, [C.POINTER(HeadWeights), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "radad_fuse_head_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "radad_synth_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_void_p]),
    "radad_synth_audio": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_int, C.c_void_p]),
}

_lib = None


class RadadHipError(RuntimeError):
    """A HIP / allocation / IO failure inside libradad_hip.so."""


def load():
    """Load libradad_hip.so once.  torch is imported first so that the HIP runtime the library binds to
    (DT_NEEDED libamdhip64.so.7) is the copy PyTorch-ROCm already mapped -- one runtime per process, so
    tensor.data_ptr() addresses and torch streams are valid inside the library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C radad_retrievalaugmenteddeepfakeaudiodetection_amd/csrc`. There is no CPU fallback.")
    import torch  # noqa: F401  (maps libamdhip64 first; see docstring)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here == the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.radad_abi_version() != 1:
        raise ImportError(f"libradad_hip.so ABI {lib.radad_abi_version()} != 1")
    _lib = lib
    return lib


def check(rc, what=""):
    """Turn a status code into the exception the reference's callers expect: ValueError for argument /
    state errors (vector_database.py:72,160-161; segmenter.py:19), RadadHipError otherwise."""
    if rc == RADAD_OK:
        return
    msg = load().radad_last_error().decode("utf-8", "replace")
    if what:
        msg = f"{what}: {msg}"
    if rc in (RADAD_EINVAL, RADAD_ESTATE):
        raise ValueError(msg)
    if rc == RADAD_ENOMEM:
        raise MemoryError(msg)
    if rc == RADAD_EIO:
        raise OSError(msg)
    raise RadadHipError(msg)


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` as a void*."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, name):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: the HIP path has no CPU fallback")
    return t
