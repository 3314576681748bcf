#!/usr/bin/env python3
"""bench.py -- clips/sec of the segment -> embed -> retrieve hot path on N MI355X of one node.

Workload (BASELINE.json metric: "clips/sec (segment+embed+retrieve) @1M x 512 DB"):
  per GPU and step: 1024 synthetic 4 s / 16 kHz clips (3 segments each), resident in HBM ->
  k_logmel + k_proj_pool (F = 512, pyramid levels [1] => D = 512) -> cosine top-10 against a
  1 000 000 x 512 fp32 reference store.  With N > 1 ranks the store is row-sharded (1M/N rows each), every
  rank embeds its own 1024 clips, the embeddings are all-gathered (RCCL), each rank scores ALL N*1024
  queries against its shard, the per-shard top-10 lists are all-gathered and merged.  Per-GPU work is
  therefore constant in N ("weak"); value = N*1024 clips / max-over-ranks step time.

One JSON line on rank 0 (contract in the task statement), including
  roofline     : the dominant kernel (the scan: k_knn_wide on the f16 matrix pipe, or k_knn_f32_reg with --scan f32)
                 timed with HIP events on the launch stream
  cpu_baseline : the oracle (numpy port of the same pipeline) timed on this host on a bounded sample,
                 which doubles as the full-size parity check of the GPU result (ids bit-exact on the sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CLIPS_PER_GPU = 1024
CLIP_SAMPLES = 64000
DB_ROWS = 1_000_000
DIM = 512
TOP_K = 10
AUDIO_SEED, DB_SEED, NOISE_SEED = 1234, 4321, 99
PEAK_MFMA_F32_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA, dense
PEAK_MFMA_F16_TFLOPS = 2500.0     # same guide: BF16/F16 MFMA ~2.5 PF dense (at 2.4 GHz; MFMA-dense loops hold 1.5-1.95 GHz)
PEAK_HBM_GBPS = 8000.0
# HBM-side bytes of ONE scan launch of the default 1-GPU workload, from the PMC passes of the same command
# (profiles/r1_b_pmc_summary.txt, tools/profile_r1.sh): FETCH_SIZE 4.184e6 KB x 1024 x 2 (gfx950 reports half of a
# 16-B/lane stream, MI355X_MICROARCH.md "HBM") + WRITE_SIZE 8.2e3 KB x 1024.  Infinity-Cache hits are counted in
# FETCH_SIZE, so this is an upper bound on DRAM traffic; it is only meaningful for that exact workload.
SCAN_TRAFFIC_BYTES_R1B = 4.18399e6 * 1024 * 2 + 8195.59 * 1024
# the same for k_knn_wide (profiles/r1_d_pmc_summary.txt): FETCH_SIZE 1.587e6 KB, WRITE_SIZE 4.83e4 KB per full-scan launch
SCAN_TRAFFIC_BYTES_WIDE = 1.587e6 * 1024 * 2 + 4.831e4 * 1024


def planted_row(j, c, n_total):
    """global row that holds near-duplicate c of global query j"""
    return (j * 977 + c * 350003 + 17) % n_total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU, help="clips per GPU per step")
    ap.add_argument("--db-rows", type=int, default=DB_ROWS, help="total reference-store rows")
    ap.add_argument("--cpu-sample", type=int, default=192, help="clips in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--workload", choices=["fixed", "ragged"], default="fixed",
                    help="fixed = 4 s clips (the headline); ragged = BASELINE config 3: release_in_the_wild-shaped variable-length "
                         "clips (log-normal, mean ~4.3 s, clipped to [0.5, 20] s) cut by the segmenter rule")
    ap.add_argument("--scan", choices=["auto", "f32"], default="auto",
                    help="auto = large batches scan on the f16 matrix pipe (split-f16 copy of the fp32 store, 3 MFMAs per fp32 "
                         "product, float64 re-rank from the fp32 rows); f32 = the fp32-MFMA tile kernel (RADAD_KNN_SPLIT=0)")
    ap.add_argument("--store-dtype", choices=["f32", "f16"], default="f32",
                    help="f16 = the reference's use_float16 knob (fp16 rows, fp16 MFMA scan); NOT the headline configuration")
    args = ap.parse_args()
    if args.scan == "f32":
        os.environ["RADAD_KNN_HI"] = "0"       # read by radad_knn_create

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # RADAD_BENCH_REHEARSE=1: rehearse the N-rank code path on ONE GPU (all ranks on cuda:0, gloo collectives staged
    # through the host).  Never used for a reported number.
    rehearse = os.environ.get("RADAD_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import ShardedSearch, shard_bounds
    lib = _lib.load()

    cfg = R.Config()
    cfg.update(device=dev, tpp_levels=[1], tpp_pooling_type="max", feature_dim=DIM, vector_db_index_type="IP",
               use_float16=(args.store_dtype == "f16"))
    fe = R.MelProjectionFeatureExtractor(cfg)
    B = args.clips
    n_total = args.db_rows
    lo, hi = shard_bounds(n_total, world, rank)

    # ---- inputs, resident in HBM before the timed region -------------------------------------------------
    if args.workload == "fixed":
        wave = torch.empty(B * CLIP_SAMPLES, device=dev, dtype=torch.float32)
        _lib.check(lib.radad_synth_audio(wave.data_ptr(), rank * B, B, CLIP_SAMPLES, AUDIO_SEED, local_rank, _lib.stream_ptr(dev)))
        offsets = np.arange(B + 1, dtype=np.int64) * CLIP_SAMPLES
        n_segments = 3 * B
    else:
        # variable-length clips: every clip is generated at 20 s and cut to its own length (device-side slicing)
        rng = np.random.default_rng(1235 + rank)
        lens = np.clip(np.exp(rng.normal(np.log(3.6), 0.6, B)), 0.5, 20.0)          # seconds; mean ~4.3
        lens = (lens * 16000).astype(np.int64)
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        wave = torch.empty(int(offsets[-1]), device=dev, dtype=torch.float32)
        full = torch.empty(320000, device=dev, dtype=torch.float32)
        for b in range(B):
            _lib.check(lib.radad_synth_audio(full.data_ptr(), rank * B + b, 1, 320000, AUDIO_SEED + 1, local_rank, _lib.stream_ptr(dev)))
            wave[offsets[b]:offsets[b + 1]] = full[:lens[b]]
        del full
        n_segments = int(sum(max(1, (int(n) - 32000) // 16000 + 1) for n in lens))
    emb0 = fe.embed_clips(wave, offsets)                              # also the first warm-up of the embed kernels
    gather = ShardedSearch(None, 0)._all_gather
    all_emb = gather(emb0) if world > 1 else emb0
    rows = torch.empty((hi - lo, DIM), device=dev, dtype=torch.float32)
    _lib.check(lib.radad_synth_rows(rows.data_ptr(), lo, hi - lo, DIM, DB_SEED, local_rank, _lib.stream_ptr(dev)))
    # plant two near-duplicates of every query so that the top of each list is known and non-trivial
    Q = world * B
    noise = torch.empty((2 * Q, DIM), device=dev)
    _lib.check(lib.radad_synth_rows(noise.data_ptr(), 0, 2 * Q, DIM, NOISE_SEED, local_rank, _lib.stream_ptr(dev)))
    jj = torch.arange(Q, device=dev)
    scale = all_emb.norm(dim=1, keepdim=True) / (DIM ** 0.5)
    for c, eps in ((0, 0.05), (1, 0.10)):
        g = (jj * 977 + c * 350003 + 17) % n_total
        mine = (g >= lo) & (g < hi)
        rows[g[mine] - lo] = all_emb[mine] + eps * scale[mine] * noise[c * Q:(c + 1) * Q][mine]
    vdb = R.VectorDatabase(cfg)
    vdb.create_index(DIM, id_base=lo)                                 # cosine: rows are normalised by the add kernel
    vdb.index.reserve(hi - lo)
    vdb.index.add_device(rows)
    torch.cuda.synchronize()
    del noise
    def local_search(q, k):       # float64 keys travel between shards; one GPU needs only the fp32 distances
        if world == 1:
            return vdb.index.search_device(q, k)
        _, ids, key64 = vdb.index.search_device(q, k, return_f64=True)
        return key64, ids
    searcher = ShardedSearch(local_search, vdb.index.metric)

    def step():
        emb = fe.embed_clips(wave, offsets)
        return emb, searcher.search(emb, TOP_K)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    fe.profile(True)
    vdb.index.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        emb, (D, I) = step()
    barrier()
    dt = time.perf_counter() - t0
    knn_ms = vdb.index.profile_read()
    lm_ms, pp_ms = fe.profile_read()
    fe.profile(False)
    vdb.index.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- correctness of the timed result (cheap, every rank): the planted rows lead every list ----------------
    mine = torch.arange(rank * B, (rank + 1) * B, device=dev)
    want0 = (mine * 977 + 17) % n_total
    planted_ok = bool((I[:, 0] == want0).all().item())
    if world > 1:                                   # rank 0 reports the verdict of ALL ranks
        t = torch.tensor([1.0 if planted_ok else 0.0], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        planted_ok = bool(t.item() > 0.5)

    ms_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    knn_avg = float(np.mean(knn_ms)) if knn_ms else float("nan")
    flops = 2.0 * Q * (hi - lo) * DIM                                   # algorithmic FLOPs of one scan launch
    alg_bytes = 4.0 * (hi - lo) * DIM + 4.0 * Q * DIM + 12.0 * Q * TOP_K
    achieved = flops / (knn_avg * 1e-3) / 1e12
    f16 = args.store_dtype == "f16"
    launch = vdb.index.last_launch()
    wide = launch["block_threads"] == 512                     # k_knn_wide (knn_wide.inc) took the scan
    if f16:
        alg_bytes = 2.0 * (hi - lo) * DIM + 2.0 * Q * DIM + 12.0 * Q * TOP_K
    if wide:
        kname = "k_knn_hi<0>"
        peak = PEAK_MFMA_F16_TFLOPS
        issued = flops                                        # one f16 MFMA product per element (certified filter)
        dtype = ("f32 embed; f16 store + f16 MFMA scan (f32 accumulate, f64 re-rank)" if f16 else
                 "f32 (scan products as 3 f16 MFMAs on hi/lo splits of the fp32 values, f32 accumulate; f64 re-rank from the fp32 rows)")
    else:
        kname = "k_knn_f32_reg<16,%s>" % ("true" if f16 else "false")
        peak = PEAK_MFMA_F16_TFLOPS if f16 else PEAK_MFMA_F32_TFLOPS
        issued = flops
        dtype = "f32" if not f16 else "f32 embed; f16 store + f16 MFMA scan (f32 accumulate, f64 re-rank)"
    traffic = None
    if world == 1 and B == CLIPS_PER_GPU and n_total == DB_ROWS and not f16 and args.workload == "fixed":
        traffic = SCAN_TRAFFIC_BYTES_WIDE if wide else SCAN_TRAFFIC_BYTES_R1B
    out = {
        "metric": "clips/sec (segment+embed+retrieve) @1Mx512 DB",
        "value": round(value, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": (f"{B} clips/GPU x 4 s @16 kHz (3 segments)" if args.workload == "fixed" else
                                f"{B} variable-length clips/GPU (log-normal, mean {float(np.mean(np.diff(offsets))) / 16000:.2f} s, "
                                f"{n_segments} segments)") +
                               f", F=512, levels=[1], cosine top-{TOP_K}, {n_total} x {DIM} {args.store_dtype} store "
                               f"row-sharded over {world} GPU(s)", "segments_per_gpu": n_segments,
                   "clips_per_gpu": B, "db_rows": n_total, "dim": DIM, "k": TOP_K, "parallelism": f"shard{world}",
                   "planted_neighbours_found": planted_ok},
        "roofline": {"kernel": kname, "bound": "mfma", "achieved": round(achieved, 2),
                     "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic,
                     "mfma_flops_issued_per_launch": issued, "issued_frac": round(issued / (knn_avg * 1e-3) / 1e12 / peak, 4),
                     "kernel_ms": round(knn_avg, 4), "flops_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes,
                     "hbm_GBps_algorithmic": round(alg_bytes / (knn_avg * 1e-3) / 1e9, 1),
                     "launch": launch},
        "kernels_ms": {"k_logmel": round(float(np.mean(lm_ms)), 4) if lm_ms else None,
                       "k_proj_pool": round(float(np.mean(pp_ms)), 4) if pp_ms else None, "scan": round(knn_avg, 4)},
    }

    # ---- CPU baseline = the oracle on this host, bounded sample; also the full-size parity check ------------------
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import radad_oracle as O
        ns = min(args.cpu_sample, B)
        wav_h = [wave[offsets[b]:offsets[b + 1]].cpu().numpy() for b in range(ns)]
        db_h = rows.cpu().numpy()
        cores = len(os.sched_getaffinity(0))
        t0 = time.perf_counter()
        emb_ref = O.embed_clips(wav_h, fe.segment_length, fe.hop_length, fe.proj_w, fe.proj_b, (1,), "max")
        t_embed = time.perf_counter() - t0
        od, oi = O.knn(db_h, emb_ref, TOP_K, "COSINE", chunk=65536)
        t_cpu = time.perf_counter() - t0
        emb_gpu = emb[:ns].cpu().numpy()
        emb_err = float(np.abs(emb_gpu - emb_ref).max())
        I_h, D_h = I[:ns].cpu().numpy(), D[:ns].cpu().numpy()
        recall = float(np.mean([len(set(a) & set(b)) / TOP_K for a, b in zip(I_h, oi)]))
        # retrieve parity proper (untimed): the reference normalises in float32 BEFORE the index sees the vectors
        # (vector_database.py:103-104,118,166), so the search is judged on the rows and queries as stored --
        # float64 inner products of exactly those float32 vectors, (distance, id) order.
        qn = torch.empty_like(emb[:ns])
        _lib.check(lib.radad_rownorm(emb[:ns].contiguous().data_ptr(), qn.data_ptr(), ns, DIM, local_rank, _lib.stream_ptr(dev)))
        stored = np.empty((hi - lo, DIM), np.float32)             # the rows exactly as the store holds them (decoded)
        for r0 in range(0, hi - lo, 131072):
            ids = torch.arange(lo + r0, min(hi, lo + r0 + 131072), device=dev)
            stored[r0:r0 + len(ids)] = vdb.index.reconstruct_batch(ids).cpu().numpy()
        sd, si = O.knn(stored, qn.cpu().numpy(), TOP_K, "IP", chunk=65536)
        del stored
        ids_equal = bool(np.array_equal(I_h, si))
        dist_err = float(np.abs(D_h - sd).max())
        norm_err = float(np.abs(qn.cpu().numpy().astype(np.float64) - O.maybe_normalize(emb_gpu, True)).max())
        out["cpu_baseline"] = {"value": round(ns / t_cpu, 2), "unit": "clips/s", "cores": cores, "kind": "port",
                               "sample": f"{ns} of the {B} clips against the full {n_total} x {DIM} store "
                                         f"(numpy float64 oracle: embed {t_embed:.1f} s + kNN {t_cpu - t_embed:.1f} s)",
                               "parity_on_sample": {"ids_bit_exact": ids_equal, "max_abs_dist_err": dist_err,
                                                    "max_abs_embed_err": emb_err, "max_abs_rownorm_err": norm_err,
                                                    "recall_at_k_vs_float64_cosine_of_oracle_embeddings": recall}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
