#!/usr/bin/env python3
"""bench.py -- clips/sec of the segment -> embed -> retrieve hot path on N MI355X of one node.

Workload (BASELINE.json metric: "clips/sec (segment+embed+retrieve) @1M x 512 DB"):
  per GPU and step: 1024 synthetic 4 s / 16 kHz clips (3 segments each), resident in HBM ->
  k_logmel_h + k_proj_pool (F = 512, pyramid levels [1] => D = 512) -> cosine top-10 against a
  1 000 000 x 512 fp32 reference store.  With N > 1 ranks the store is row-sharded (1M/N rows each), every
  rank embeds its own 1024 clips, the embeddings are all-gathered (RCCL), each rank scores ALL N*1024
  queries against its shard, the per-shard top-10 lists are exchanged (all-to-all) and merged.  Per-GPU work is
  therefore constant in N ("weak"); value = N*1024 clips / max-over-ranks step time.  The store is the same 1 M rows for every N
  (random rows + two near-duplicates of each of rank 0's 1024 clips).

One JSON line on rank 0 (contract in the task statement), including
  roofline     : the dominant kernel (the scan: k_knn_hi on the f16 matrix pipe, or k_knn_f32_reg with --scan f32), timed
                 with HIP events on the launch stream over the timed region; `kernels` adds the same accounting for the log-mel and
                 projection kernels, from event pairs over as many steps right after it
  sustained    : the same step looped for >= --sustain seconds after the timed steps (thermal steady state)
  cpu_baseline : the float32 torch-CPU port of the same pipeline (FFT + BLAS, all host cores; oracle/cpu_baseline.py)
                 timed on this host on a bounded sample; the float64 oracle is the CHECKER of the GPU result
                 (`parity_on_sample`: ids bit-exact), never the thing timed.
--workload ragged : BASELINE config 3 (variable-length clips); two batches with different offsets alternate, so the
                 segment plan (built on the device) is rebuilt inside the timed region every step.
--mode predict : the online case (pipeline.py:1038-1054): --predict-queries (<= 16) queries per search, HBM-bound
                 streaming scan; reports searches/s and the scan's HBM GB/s against the 8 TB/s peak.
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
PEAK_VALU_F32_TFLOPS = 157.3      # same guide: fp32 vector peak (reached with packed fp32: 256 CUs x 4 SIMDs x 16 lanes x 2 x 2 x 2.4 GHz)


def scan_traffic_from_profiles(section):
    """HBM-side bytes of ONE scan launch of the default 1-GPU workload: FETCH_SIZE (KB) x 2 (gfx950 reports half of a 16-B/lane
    stream, MI355X_MICROARCH.md "HBM") + WRITE_SIZE (KB), read from the newest profiles/*_pmc_summary.txt (the PMC passes of this
    same command, tools/profile_r1.sh; RADAD_PMC_SUMMARY names another file).  PMC counters cannot be collected inside the run
    that prints the line (separate rocprofv3 passes), so the figure is only as fresh as that file: its name travels with the
    number in `traffic_source`.  Infinity-Cache hits are counted in FETCH_SIZE: an upper bound on DRAM traffic."""
    import glob
    import re
    def round_key(path):        # r3_d_pmc_summary.txt -> (3, "d"): by round NUMBER, then letter (a plain sort puts r10 before r3)
        m = re.match(r"r(\d+)_([a-z0-9]*)_", os.path.basename(path))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    cand = [os.environ["RADAD_PMC_SUMMARY"]] if os.environ.get("RADAD_PMC_SUMMARY") else \
        sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")), key=round_key)
    for path in reversed(cand):
        try:
            text = open(path).read()
        except OSError:
            continue
        vals, cur = {}, None
        for line in text.splitlines():
            if line and not line[0].isspace() and not line.startswith("#"):
                cur = line.strip()
            elif cur == section:
                m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+mean\s+([0-9.eE+-]+)", line)
                if m:
                    vals[m.group(1)] = float(m.group(2))
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            return vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024, {
                "file": os.path.relpath(path, ROOT), "kernel_section": section, "fetch_size_kb": vals["FETCH_SIZE"],
                "write_size_kb": vals["WRITE_SIZE"], "formula": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024",
                "measured_in_this_run": False}
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (0.2 s at the default workload)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--clips", type=int, default=CLIPS_PER_GPU, help="clips per GPU per step")
    ap.add_argument("--db-rows", type=int, default=DB_ROWS, help="total reference-store rows")
    ap.add_argument("--cpu-sample", type=int, default=192, help="clips checked against the float64 oracle (0 = skip the check)")
    ap.add_argument("--cpu-baseline-clips", type=int, default=512, help="clips timed on the torch-CPU baseline (0 = skip)")
    ap.add_argument("--sustain", type=float, default=5.0, help="seconds of the same step looped after the timed steps (0 = skip)")
    ap.add_argument("--pcie", type=float, default=2.0, help="seconds of the step looped with every batch's audio crossing PCIe "
                    "(pinned host -> device, double-buffered on a copy stream, overlapped with the previous step); 0 = skip")
    ap.add_argument("--unstructured", type=int, default=10, help="searches of 1024 random (unstructured) queries timed after the "
                    "headline steps for the `scan_unstructured_ms` key (0 = skip)")
    ap.add_argument("--shard-bound", type=int, default=0, help="N > 1, --parallelism shard: 1 = exchange per-query k-th-best bounds between "
                    "scan and re-rank (pays on stores whose queries have distinct neighbourhoods; on this benchmark's data, where every "
                    "planted row neighbours every query, it costs 2.5-8.8 %%: profiles/r3_e_rehearse.txt), 0 (default) = every shard "
                    "certifies its own top k")
    ap.add_argument("--parallelism", choices=["shard", "replicate"], default="shard",
                    help="N > 1: shard (default, north_star) = the store is row-sharded, every rank scans all queries against its "
                         "shard, the per-shard lists are exchanged and merged; replicate = every rank holds the WHOLE store and searches "
                         "only its own clips: no collective at all (any BASELINE store fits one 288 GB GPU)")
    ap.add_argument("--mode", choices=["step", "predict"], default="step")
    ap.add_argument("--predict-queries", type=int, default=1)
    ap.add_argument("--end-to-end", action="store_true", help="--mode predict: time HotPathPipeline.predict (pipeline.py:1038-1103) for one "
                    "3 s clip -- load -> embed -> search(exclude_self) -> RADADModel -> sigmoid -- against a --db-rows x 3584 L2 store "
                    "(F = 512, levels [1, 2, 4]: the reference's Whisper shape); latency per call, host work included")
    ap.add_argument("--dim", type=int, default=DIM, help="--mode predict only: embedding dimension of the store (the reference's own "
                    "stores are 5376- or 3584-dimensional: 7 pyramid bins x 768 / 512 features); != 512 uses synthetic queries")
    ap.add_argument("--metric", choices=["cosine", "l2"], default="cosine", help="--mode predict only (the reference's default index is L2)")
    ap.add_argument("--k", type=int, default=TOP_K, help="--mode predict only (the reference searches K + 10 = 15)")
    ap.add_argument("--workload", choices=["fixed", "ragged"], default="fixed",
                    help="fixed = 4 s clips (the headline); ragged = BASELINE config 3: release_in_the_wild-shaped variable-length "
                         "clips (log-normal, mean ~4.3 s, clipped to [0.5, 20] s) cut by the segmenter rule")
    ap.add_argument("--scan", choices=["auto", "f32"], default="auto",
                    help="auto = large batches take the certified single-product f16 scan on the hi plane of the fp32 store "
                         "(float64 re-rank from the fp32 rows); f32 = the fp32-MFMA tile kernel only (config.knn_hi_plane = False)")
    ap.add_argument("--logmel", choices=["fft", "gemm"], default="fft",
                    help="fft = the shared-frame log-mel as a radix FFT on the vector ALU (k_logmel_fft_clip, default); gemm = the round-3 "
                         "DFT-as-GEMM on the f16 matrix pipe (k_logmel_h_clip), for A/B")
    ap.add_argument("--live-floor", type=int, choices=[0, 1], default=0,
                    help="0 (default) = the tile scan runs one launch per phase with k_kth_floor between them; 1 = ONE launch that raises its "
                         "admission floors inside it (round 5; measured 1-4 %% slower: profiles/r5_ab_*.txt), for A/B")
    ap.add_argument("--store-dtype", choices=["f32", "f16"], default="f32",
                    help="f16 = the reference's use_float16 knob (fp16 rows); NOT the headline configuration")
    ap.add_argument("--embed-dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 = embeddings emitted and searched as bfloat16 (BASELINE config 5); NOT the headline configuration")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started bare as `python bench.py --gpus N`: become N ranks.  The parent only COUNTS devices (torch.cuda.device_count() does
        # not initialise the GPU on this image), spawns `python -m torch.distributed.run` as a CHILD process (nothing that has touched
        # the GPU is ever exec'd), lets rank 0's JSON line through on the inherited stdout and exits with the launcher's status.
        # Fewer than N devices is an error, never a silent one-rank run -- unless RADAD_BENCH_REHEARSE=1 (all ranks on cuda:0,
        # gloo collectives: the N-rank code path on a one-GPU box, never a reported number).
        import socket
        import subprocess
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and os.environ.get("RADAD_BENCH_REHEARSE", "0") != "1":
            print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible; refusing to fall back to fewer ranks "
                  f"(RADAD_BENCH_REHEARSE=1 rehearses the {args.gpus}-rank path on one GPU over gloo)", file=sys.stderr, flush=True)
            sys.exit(2)
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr, flush=True)
        sys.exit(2)
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
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd.sharded import ReplicatedSearch, ShardedSearch, shard_bounds
    lib = _lib.load()

    if args.mode == "predict" and args.end_to_end:
        assert world == 1, "--mode predict is a one-GPU measurement"
        cfg = R.Config()
        cfg.update(device=dev, feature_dim=512, tpp_levels=[1, 2, 4], top_k=5, vector_db_index_type="L2",
                   vector_db_path=os.path.join("/tmp", f"radad_bench_vdb_{os.getpid()}"))
        pipe = R.HotPathPipeline(cfg)
        D_ = pipe.tpp.get_output_dim()
        n_rows = args.db_rows if args.db_rows != DB_ROWS else 25423
        clip = torch.empty(48000, device=dev, dtype=torch.float32)
        _lib.check(lib.radad_synth_audio(clip.data_ptr(), 0, 1, 48000, AUDIO_SEED, local_rank, _lib.stream_ptr(dev)))
        clip_h = clip.cpu().numpy()

        class _DS:
            def load_audio(self, path):
                return clip_h
        emb = pipe.process_audio_batch(["/eval/q.wav"], _DS())
        rows = torch.empty((n_rows, D_), device=dev, dtype=torch.float32)
        _lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, n_rows, D_, DB_SEED, local_rank, _lib.stream_ptr(dev)))
        rows = rows * (0.05 * emb.abs().mean()) + emb.mean()
        rows[17] = emb[0] + 0.01 * rows[18]
        pipe.vector_db.add_vectors(rows, [f"/train/f{i}.wav" for i in range(n_rows)], [float(i % 2) for i in range(n_rows)],
                                   {"speaker_id": ["s"] * n_rows})
        model = R.RADADModel(cfg, D_).eval().to(dev)
        for _ in range(max(3, args.warmup)):
            out_p = pipe.predict("/eval/q.wav", _DS(), model)
        torch.cuda.synchronize()
        lat = []
        for _ in range(args.steps):
            t0 = time.perf_counter()
            out_p = pipe.predict("/eval/q.wav", _DS(), model)
            lat.append(1e3 * (time.perf_counter() - t0))
        # the device part alone: the same calls queued without reading anything back
        t0 = time.perf_counter()
        for _ in range(args.steps):
            e_ = pipe.process_audio_batch(["/eval/q.wav"], _DS())
            v_, l_ = pipe.retrieve_similar_vectors(e_, query_paths=["/eval/q.wav"], exclude_self=True)
            lg_ = model(v_, e_)
        torch.cuda.synchronize()
        dev_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        lat = np.asarray(lat)
        print(json.dumps({"metric": f"predict() latency, one 3 s clip @{n_rows}x{D_} L2 store (pipeline.py:1038-1103)",
                          "value": round(float(np.median(lat)), 4), "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(lat.mean()), 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"HotPathPipeline.predict: load -> embed (F=512, levels [1,2,4]) -> L2 top-15 with the "
                                                 f"clip's basename excluded -> RADADModel -> sigmoid; {n_rows} x {D_} store", "db_rows": n_rows,
                                     "dim": D_, "k_search": 15},
                          "latency_ms": {"median": round(float(np.median(lat)), 4), "p90": round(float(np.percentile(lat, 90)), 4),
                                         "min": round(float(lat.min()), 4)},
                          "queued_without_readback_ms": round(dev_ms, 4),
                          "result": {k: out_p[k] for k in ("prediction", "probability_spoof", "retrieved_files")},
                          "scan_kind": pipe.vector_db.index.last_launch()["scan_kind"]}), flush=True)
        return

    if args.mode == "predict" and (args.dim != DIM or args.metric != "cosine" or args.k != TOP_K):
        # ---- the online search at the reference's own shape (pipeline.py:1038-1054: ONE query, D = 5376 / 3584, k = 15, L2 by
        # default; config.py:48-56): synthetic embeddings (there is no D-dimensional encoder here), one planted neighbour per query
        assert world == 1, "--mode predict is a one-GPU measurement"
        dim, n_rows, k = args.dim, args.db_rows, args.k
        rows = torch.empty((n_rows, dim), device=dev, dtype=torch.float32)
        _lib.check(lib.radad_synth_rows(rows.data_ptr(), 0, n_rows, dim, DB_SEED, local_rank, _lib.stream_ptr(dev)))
        nqp = max(1, min(16, args.predict_queries))
        qp = torch.empty((nqp, dim), device=dev, dtype=torch.float32)
        _lib.check(lib.radad_synth_rows(qp.data_ptr(), 0, nqp, dim, 977, local_rank, _lib.stream_ptr(dev)))
        planted = (torch.arange(nqp, device=dev) * 977 + 17) % n_rows
        rows[planted] = qp + 0.05 * rows[:nqp]
        index = R.HipFlatIndex(dim, _lib.METRIC_L2 if args.metric == "l2" else _lib.METRIC_COSINE, local_rank,
                               store_f16=(args.store_dtype == "f16"))
        index.add_device(rows)
        for _ in range(args.warmup):
            index.search_device(qp, k)
        torch.cuda.synchronize()
        index.profile(True, every=4)      # (an event pair holds the dependent kernels back ~12 us: every 4th search carries one)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            D, I = index.search_device(qp, k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        scan = float(np.mean(index.profile_read()))
        launch = index.last_launch()
        esz = 2.0 if (args.store_dtype == "f16" or launch["scan_kind"] in ("hi_smallq", "hi_tile")) else 4.0
        byts = esz * n_rows * dim + 4.0 * nqp * dim + 12.0 * nqp * k
        out = {"metric": f"searches/sec (retrieve, online predict path) @{n_rows}x{dim} DB", "value": round(args.steps / dt, 1),
               "unit": "searches/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{nqp} query(ies) per search, {args.metric} top-{k}, {n_rows} x {dim} {args.store_dtype} store",
                          "queries_per_search": nqp, "db_rows": n_rows, "dim": dim, "k": k},
               "roofline": {"kernel": launch["scan_kind"], "bound": "hbm", "achieved": round(byts / (scan * 1e-3) / 1e9, 1),
                            "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(byts / (scan * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                            "traffic": None, "kernel_ms": round(scan, 4), "algorithmic_bytes_per_launch": byts, "launch": launch},
               "planted_neighbours_found": bool((I[:, 0] == planted).all().item())}
        print(json.dumps(out), flush=True)
        return

    cfg = R.Config()
    cfg.update(device=dev, tpp_levels=[1], tpp_pooling_type="max", feature_dim=DIM, vector_db_index_type="IP",
               use_float16=(args.store_dtype == "f16"), knn_hi_plane=(False if args.scan == "f32" else None),
               melproj_logmel_fft=(args.logmel == "fft"), knn_live_floor=(args.live_floor if args.live_floor else None))
    fe = R.MelProjectionFeatureExtractor(cfg)
    B = args.clips
    n_total = args.db_rows
    f16_store = args.store_dtype == "f16"
    replicate = world > 1 and args.parallelism == "replicate"
    lo, hi = (0, n_total) if replicate else shard_bounds(n_total, world, rank)
    emb_dtype = torch.bfloat16 if args.embed_dtype == "bf16" else torch.float32

    # ---- inputs, resident in HBM before the timed region -------------------------------------------------
    # batches[i] = (wave, clip offsets as handed to embed_clips, segments, host offsets); the fixed workload has one batch,
    # the ragged one alternates two
    batches = []
    if args.workload == "fixed":
        wave = torch.empty(B * CLIP_SAMPLES, device=dev, dtype=torch.float32)
        _lib.check(lib.radad_synth_audio(wave.data_ptr(), rank * B, B, CLIP_SAMPLES, AUDIO_SEED, local_rank, _lib.stream_ptr(dev)))
        offs = np.arange(B + 1, dtype=np.int64) * CLIP_SAMPLES
        batches.append((wave, offs, 3 * B, offs))
    else:
        # variable-length clips: every clip is generated at 20 s and cut to its own length (device-side slicing); the offsets
        # live on the DEVICE (radad_embed_forward_dev builds the segment plan there: nothing synchronises with the host)
        full = torch.empty(320000, device=dev, dtype=torch.float32)
        for bi in range(2):
            rng = np.random.default_rng(1235 + 1000 * bi + rank)
            lens = np.clip(np.exp(rng.normal(np.log(3.6), 0.6, B)), 0.5, 20.0)          # seconds; mean ~4.3
            lens = (lens * 16000).astype(np.int64)
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            wave = torch.empty(int(offs[-1]), device=dev, dtype=torch.float32)
            for b in range(B):
                _lib.check(lib.radad_synth_audio(full.data_ptr(), (2 * rank + bi) * B + b, 1, 320000, AUDIO_SEED + 1, local_rank,
                                                 _lib.stream_ptr(dev)))
                wave[offs[b]:offs[b + 1]] = full[:lens[b]]
            n_seg = int(sum(max(1, (int(n) - 32000) // 16000 + 1) for n in lens))
            batches.append((wave, torch.from_numpy(offs).to(dev), n_seg, offs))
        del full
    n_segments = batches[0][2]

    def embed(i):
        w, o = batches[i % len(batches)][:2]
        return fe.embed_clips(w, o, out_dtype=emb_dtype)

    emb0 = embed(0).float()                                           # also the first warm-up of the embed kernels
    gather = ShardedSearch(None, 0)._all_gather
    all_emb = gather(emb0) if world > 1 else emb0
    rows = torch.empty((hi - lo, DIM), device=dev, dtype=torch.float32)
    _lib.check(lib.radad_synth_rows(rows.data_ptr(), lo, hi - lo, DIM, DB_SEED, local_rank, _lib.stream_ptr(dev)))
    # plant two near-duplicates of every clip of RANK 0's batch 0 (2 B rows) so that the top of each list is known and non-trivial.
    # The store is the SAME 1 M rows whatever the number of GPUs that search it (until round 5 every rank's queries were planted:
    # synthetic clips embed within cos 0.99 of each other, so every planted row neighbours every query, and a store whose planted
    # rows grow with N gave every rank N x the candidates to re-rank -- per-GPU work that was not constant in N: the rehearsal's
    # 0.79 of linear at N = 8 was the generator's, not the layout's).
    Q = world * B                                                     # (queries a rank SCANS: all of them when sharded)
    noise = torch.empty((2 * B, DIM), device=dev)
    _lib.check(lib.radad_synth_rows(noise.data_ptr(), 0, 2 * B, DIM, NOISE_SEED, local_rank, _lib.stream_ptr(dev)))
    jj = torch.arange(B, device=dev)
    emb_r0 = all_emb[:B]
    scale = emb_r0.norm(dim=1, keepdim=True) / (DIM ** 0.5)
    planted_ids = []
    for c, eps in ((0, 0.05), (1, 0.10)):
        g = (jj * 977 + c * 350003 + 17) % n_total
        planted_ids.append(g)
        mine = (g >= lo) & (g < hi)
        rows[g[mine] - lo] = emb_r0[mine] + eps * scale[mine] * noise[c * B:(c + 1) * B][mine]
    planted_ids = torch.cat(planted_ids)
    vdb = R.VectorDatabase(cfg)
    vdb.create_index(DIM, id_base=lo)                                 # cosine: rows are normalised by the add kernel (replicate: lo = 0)
    vdb.index.reserve(hi - lo)
    vdb.index.add_device(rows)
    torch.cuda.synchronize()
    del noise

    def local_search(q, k):       # float64 keys travel between shards; one GPU needs only the fp32 distances
        if world == 1:
            return vdb.index.search_device(q, k)
        _, ids, key64 = vdb.index.search_device(q, k, return_f64=True)
        return key64, ids
    # N > 1: the search runs in two halves around an all-reduce(max) of per-query bounds (sharded.py), so that every shard
    # re-ranks only what can be among the GLOBAL top k
    bounded = None
    if world > 1 and args.shard_bound and not replicate:
        def _finish(lb):
            _, ids, key64 = vdb.index.search_finish(lb, return_f64=True)
            return key64, ids
        bounded = (vdb.index.search_begin, _finish, vdb.index.search_abort)
    if replicate:                                           # the whole store on every rank: a rank's search is the one-GPU search
        searcher = ReplicatedSearch(lambda q, k: vdb.index.search_device(q, k))
    else:
        searcher = ShardedSearch(local_search, vdb.index.metric, bounded=bounded, timing=world > 1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.mode == "predict":
        # ---- the online case: a handful of queries per search, the scan streams the store once (HBM-bound) ----------
        assert world == 1, "--mode predict is a one-GPU measurement"
        nqp = max(1, min(16, args.predict_queries))
        qp = emb0[:nqp].contiguous()
        for _ in range(args.warmup):
            vdb.index.search_device(qp, TOP_K)
        barrier()
        vdb.index.profile(True, every=4)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            D, I = vdb.index.search_device(qp, TOP_K)
        barrier()
        dt = time.perf_counter() - t0
        ms = vdb.index.profile_read()
        scan = float(np.mean(ms))
        launch = vdb.index.last_launch()
        # bytes the scan kernel streams: the f16 plane of an fp32 store (or the fp16 store) on the certified small-batch kernel,
        # the fp32 rows otherwise
        esz = 2.0 if (args.store_dtype == "f16" or launch["scan_kind"] == "hi_smallq") else 4.0
        byts = esz * (hi - lo) * DIM + 4.0 * nqp * DIM + 12.0 * nqp * TOP_K
        kname = {"hi_smallq": "k_knn_hi_smallq<16>", "f32_smallq": "k_knn_f32_smallq<16>", "f16_tile": "k_knn_f32_reg<16,true>"}.get(
            launch["scan_kind"], launch["scan_kind"])
        out = {"metric": "searches/sec (retrieve, online predict path) @1Mx512 DB", "value": round(args.steps / dt, 1), "unit": "searches/s",
               "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{nqp} query(ies) per search, cosine top-{TOP_K}, {n_total} x {DIM} {args.store_dtype} store",
                          "queries_per_search": nqp, "db_rows": n_total, "dim": DIM, "k": TOP_K},
               "roofline": {"kernel": kname, "bound": "hbm",
                            "achieved": round(byts / (scan * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                            "frac": round(byts / (scan * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4), "traffic": None, "kernel_ms": round(scan, 4),
                            "algorithmic_bytes_per_launch": byts, "launch": launch},
               "planted_neighbours_found": bool((I[:, 0] == (torch.arange(nqp, device=dev) * 977 + 17) % n_total).all().item())}
        print(json.dumps(out), flush=True)
        return

    step_no = [0]

    def step():
        emb = embed(step_no[0])
        step_no[0] += 1
        return emb, searcher.search(emb, TOP_K)

    for _ in range(args.warmup):
        step()
    barrier()
    step_no[0] = 0
    # HIP events on the launch stream bracket the launches of the DOMINANT kernel (the scan) in every 4th step of the timed region:
    # an event record between dependent kernels holds the next one back ~6 us (rocprofv3 trace, profiles/r4_*_step_trace.txt: 6.0-6.5 us
    # before and after each of the step's two scan launches, 0.0 between the kernels without events), i.e. 0.024 ms per step when every
    # step carries them.  The two embedding kernels' event pairs (the `kernels` extras) are taken right AFTER the timed region.
    vdb.index.profile(True, every=4 if args.steps >= 8 else 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        emb, (D, I) = step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    knn_ms = vdb.index.profile_read()
    fe.profile(True)
    for _ in range(min(args.steps, 20)):
        step()
    barrier()
    vdb.index.profile_read()
    lm_ms, pp_ms = fe.profile_read()
    fe.profile(False)
    launch = vdb.index.last_launch()
    rechecked = launch["rechecked_queries"]
    n_launch = max(1, launch.get("scan_launches", 1))       # the tile scan covers a large store in several launches: one entry each
    shard_ms = searcher.timings() if (world > 1 and not replicate) else []
    searcher.timing = False                                   # (the legs below are not broken down: no events to create)

    # ---- sustained: the same step, looped for >= args.sustain seconds (the chip settles at its steady-state clock) ------
    sustained = None
    if args.sustain > 0:
        t1 = time.perf_counter()
        n_s = 0
        while True:
            for _ in range(25):
                step()
            n_s += 25
            barrier()
            el = max_over_ranks(time.perf_counter() - t1)
            if el >= args.sustain:
                break
        k2 = vdb.index.profile_read()
        sustained = {"seconds": round(el, 2), "steps": n_s, "value": round(world * B * n_s / el, 1), "unit": "clips/s",
                     "ms_per_step": round(1e3 * el / n_s, 4), "scan_ms": round(float(np.mean(k2)) * n_launch, 4)}
    # ---- the same scan on UNSTRUCTURED queries (random directions against the same store: no planted rows near them, so the
    # sample-based admission floor is low and the epilogue's push / drain path works hardest -- what a store of unrelated
    # embeddings looks like to the filter), cosine on the benchmark's store and L2 (the reference's default metric) on a copy
    scan_unstructured = None
    if world == 1 and args.unstructured > 0:
        qu = torch.empty((B, DIM), device=dev, dtype=torch.float32)
        _lib.check(lib.radad_synth_rows(qu.data_ptr(), 0, B, DIM, 977, local_rank, _lib.stream_ptr(dev)))
        scan_unstructured = {}
        # (the L2 copy holds the rows AS STORED by the cosine index -- unit norm: what an L2 index over normalised embeddings holds)
        l2_index = R.HipFlatIndex(DIM, _lib.METRIC_L2, local_rank, store_f16=f16_store)
        for r0 in range(0, hi - lo, 1 << 17):
            l2_index.add_device(vdb.index.reconstruct_batch(torch.arange(lo + r0, min(hi, lo + r0 + (1 << 17)), device=dev)))
        for name, index in (("cosine", vdb.index), ("l2", l2_index)):
            index.profile(False)
            for _ in range(3):
                index.search_device(qu, TOP_K)
            index.profile(True)
            for _ in range(args.unstructured):
                index.search_device(qu, TOP_K)
            ms_u = index.profile_read()
            la = index.last_launch()
            scan_unstructured[name] = {"scan_ms": round(float(np.mean(ms_u)) * max(1, la.get("scan_launches", 1)), 4), "scan_kind": la["scan_kind"],
                                       "rejected": la["certificate"]["rejected"],
                                       "candidates_per_query": round(la["certificate"]["candidates_rescored"] / B, 1)}
            index.profile(False)
        del l2_index, qu
    fe.profile(False)
    vdb.index.profile(False)

    # ---- PCIe-inclusive rate (never `value`): every step's audio starts in pinned HOST memory and is copied to one of two
    # device buffers on a copy stream while the previous step computes (what a caller that streams clips from the host gets)
    pcie = None
    pcie_pcm16 = None
    if args.pcie > 0 and args.workload == "fixed":
        def pcie_leg(pcm16):
            # pcm16: the batch crosses PCIe as 16-bit PCM (what audio files hold) and is converted on the device as the reference's
            # loader would (sample / 32768, radad_pcm16_to_f32) -- half the bytes; the timing does not depend on the values
            w_dev = batches[0][0]
            src = (w_dev * 32767.0 / max(1e-9, float(w_dev.abs().max()))).round().to(torch.int16) if pcm16 else w_dev
            w_host = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
            w_host.copy_(src)
            bufs = [torch.empty_like(src), torch.empty_like(src)]
            copy_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            ready = [torch.cuda.Event(), torch.cuda.Event()]
            free = [torch.cuda.Event(), torch.cuda.Event()]
            offs0 = batches[0][1]

            def upload(i):
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(free[i])                     # the step that read this buffer last has finished
                    bufs[i].copy_(w_host, non_blocking=True)
                    ready[i].record(copy_stream)
            for i in range(2):
                free[i].record(main)
            upload(0)
            barrier()
            t2 = time.perf_counter()
            n_p = 0
            while True:
                for _ in range(10):
                    i = n_p & 1
                    upload(i ^ 1)                                        # next step's audio crosses PCIe while this step computes
                    main.wait_event(ready[i])
                    e_ = fe.embed_clips(bufs[i], offs0, out_dtype=emb_dtype)
                    searcher.search(e_, TOP_K)
                    free[i].record(main)
                    n_p += 1
                barrier()
                el2 = max_over_ranks(time.perf_counter() - t2)
                if el2 >= args.pcie:
                    break
            gb = src.numel() * src.element_size() / 1e9
            return {"seconds": round(el2, 2), "steps": n_p, "value": round(world * B * n_p / el2, 1), "unit": "clips/s",
                    "ms_per_step": round(1e3 * el2 / n_p, 4), "h2d_GB_per_step": round(gb, 4),
                    "h2d_GBps_sustained": round(gb * n_p / el2, 1),
                    "note": ("audio H2D as 16-bit PCM, converted on the device (sample / 32768), " if pcm16 else "audio H2D ") +
                            "(pinned, copy stream, double-buffered) overlapped with compute; NOT the headline value"}
        pcie = pcie_leg(False)
        pcie_pcm16 = pcie_leg(True)

    # ---- correctness of the result (cheap, every rank): batch 0's planted rows lead every list ------------------------
    emb = embed(0)
    D, I = searcher.search(emb, TOP_K)
    # rank 0: the top hit of clip j is ITS planted row; other ranks (their clips have no planted row of their own): some planted row
    if rank == 0:
        planted_ok = bool((I[:, 0] == (torch.arange(B, device=dev) * 977 + 17) % n_total).all().item())
    else:
        planted_ok = bool(torch.isin(I[:, 0], planted_ids).all().item())
    if world > 1:                                   # rank 0 reports the verdict of ALL ranks
        t = torch.tensor([1.0 if planted_ok else 0.0], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        planted_ok = bool(t.item() > 0.5)

    ms_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    # scan time of one search = its launches added up; the roofline is quoted per LAUNCH (average duration against average work),
    # which is what `rocprofv3 --stats` averages too
    knn_launch_avg = float(np.mean(knn_ms)) if knn_ms else float("nan")
    knn_avg = knn_launch_avg * n_launch
    if replicate:
        Q = B                                                           # a replica scans only its own clips
    flops = 2.0 * Q * (hi - lo) * DIM                                   # algorithmic FLOPs of one scan launch
    f16 = args.store_dtype == "f16"
    wide = launch["block_threads"] == 512                     # the certified f16 scan (knn_hi.inc) took the search
    # bytes one SEARCH must stream: the operand the scan kernel reads -- the 2-byte f16 plane (or fp16 store) on the certified scan,
    # the 4-byte rows on the fp32 kernels -- + the queries once + the result lists (SURVEY 8d's s_db * N * D + 4 Q D + 12 Q k with
    # s_db = what is actually scanned); the fp32 store's own size is a separate key
    esz = 2.0 if (f16 or wide) else 4.0
    alg_bytes = esz * (hi - lo) * DIM + 4.0 * Q * DIM + 12.0 * Q * TOP_K
    store_bytes = (2.0 if f16 else 4.0) * (hi - lo) * DIM
    achieved = flops / (knn_avg * 1e-3) / 1e12
    if wide:
        kname = "k_knn_hi<0>"
        peak = PEAK_MFMA_F16_TFLOPS
        scan_desc = ("certified single-product f16 MFMA scan (f32 accumulate) over the " +
                     ("fp16 rows" if f16 else "f16 hi plane of the fp32 rows") +
                     "; every candidate within the error bound re-ranked in float64 from the stored rows; exact float64 "
                     "kernel for queries the certificate rejects")
    else:
        kname = "k_knn_f32_reg<16,%s>" % ("true" if f16 else "false")
        peak = PEAK_MFMA_F16_TFLOPS if f16 else PEAK_MFMA_F32_TFLOPS
        scan_desc = "fp32 MFMA scan; float64 re-rank from the stored rows"
    dtype = ("f32" if args.embed_dtype == "f32" else "bf16 embeddings") + (" / f16 store" if f16 else "") + " (" + scan_desc + ")"
    traffic, traffic_source = None, None
    if world == 1 and B == CLIPS_PER_GPU and n_total == DB_ROWS and not f16 and args.workload == "fixed" and wide:
        traffic, traffic_source = scan_traffic_from_profiles("k_knn_hi")
    logmel_kind = fe.last_logmel_kind()
    lm_avg = float(np.mean(lm_ms)) if lm_ms else float("nan")
    pp_avg = float(np.mean(pp_ms)) if pp_ms else float("nan")
    # Work of the two embedding kernels.  Algorithmic bytes: every clip sample once + the log-mel rows written
    # (log-mel); the rows read + the embeddings written (projection).  FLOPs EXECUTED depend on the kernel that ran: the radix FFT
    # (k_logmel_fft_clip: packed 200-point complex FFT ~7.6 k + real split 2 k + power 0.6 k + sparse mel 0.8 k = ~11 kFLOP per
    # transformed frame, frames that overlapping segments share transformed once) or the DFT-as-GEMM (k_logmel_h[_clip]: 201 bins x
    # 400 taps x (re, im) x 2 x 3 split products + the mel GEMM per transformed frame).
    segs_per_clip = n_segments / max(1, B)
    frames_per_seg = 200
    if logmel_kind.startswith("clip_frames"):
        n_transforms = int(B * ((segs_per_clip - 1) * 100 + frames_per_seg - 3) + 3 * n_segments)       # interior frames once per clip + 3 edge frames per segment
    else:
        n_transforms = n_segments * frames_per_seg
    if logmel_kind == "clip_frames_fft":
        lm_name, lm_flops, lm_peak, lm_bound = "k_seg_stats+k_logmel_fft_clip", n_transforms * 11000, PEAK_VALU_F32_TFLOPS, "valu"
    else:
        lm_name = "k_seg_stats+k_logmel_h_clip" if logmel_kind == "clip_frames" else "k_logmel_h"
        lm_flops, lm_peak, lm_bound = n_transforms * (201 * 400 * 2 * 2 * 3 + 201 * 80 * 2 * 3), PEAK_MFMA_F16_TFLOPS, "mfma"
    pp_flops = n_segments * (200 * 80 * DIM * 2) * 3                                 # three split-f16 products per fp32 product
    n_samples = int(batches[0][3][-1]) if args.workload != "fixed" else B * 64000
    lm_bytes = n_samples * 4 + n_segments * (200 * 80 * 4)
    pp_bytes = n_segments * (200 * 80 * 4) + B * DIM * 4

    def kroof(ms, fl, by, bound, peak):
        tf = fl / (ms * 1e-3) / 1e12
        gbs = by / (ms * 1e-3) / 1e9
        return {"kernel_ms": round(ms, 4), "bound": bound, "flops_executed_per_launch": fl, "achieved": round(tf, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(tf / peak, 4), "algorithmic_bytes_per_launch": by,
                "hbm_GBps_algorithmic": round(gbs, 1), "frac_hbm": round(gbs / PEAK_HBM_GBPS, 4)}
    if args.workload == "fixed":
        wl = f"{B} clips/GPU x 4 s @16 kHz (3 segments)"
    else:
        wl = (f"{B} variable-length clips/GPU (log-normal, mean {float(np.mean(np.diff(batches[0][3]))) / 16000:.2f} s, "
              f"{n_segments} segments; 2 batches with different offsets alternate: segment plan rebuilt on the device every step)")
    out = {
        "metric": "clips/sec (segment+embed+retrieve) @1Mx512 DB",
        "value": round(value, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": wl + f", F=512, levels=[1], cosine top-{TOP_K}, {n_total} x {DIM} {args.store_dtype} store " +
                                    (f"replicated on each of {world} GPUs" if replicate else f"row-sharded over {world} GPU(s)"), "segments_per_gpu": n_segments,
                   "clips_per_gpu": B, "db_rows": n_total, "dim": DIM, "k": TOP_K, "parallelism": (f"replicate{world}" if replicate else f"shard{world}"),
                   # fixed workload: the clip offsets do not change between steps, so radad_embed_forward reuses the segment plan
                   # (k_build_plan, 14 us, skipped); --workload ragged rebuilds it on the device every step
                   "plan_rebuilt_every_step": args.workload != "fixed",
                   "planted_neighbours_found": planted_ok},
        "roofline": {"kernel": kname, "bound": "mfma", "achieved": round(achieved, 2),
                     "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_ms": round(knn_launch_avg, 4), "launches_per_step": n_launch, "scan_ms_per_step": round(knn_avg, 4),
                     "flops_per_launch": flops / n_launch, "algorithmic_bytes_per_launch": alg_bytes / n_launch,
                     "algorithmic_bytes_per_search": alg_bytes, "scanned_bytes_per_element": esz,
                     "store_bytes_resident": store_bytes,
                     # what the launches of one search read and write, queries re-read once per chunk of the store included
                     "kernel_operand_bytes_per_launch": ((2.0 * (hi - lo) * DIM + 2.0 * Q * DIM * max(1, launch["db_splits"]) / max(1, launch["query_tiles"])
                                                          + 8.0 * Q * 330) if wide else alg_bytes) / n_launch,
                     "hbm_GBps_algorithmic": round(alg_bytes / (knn_avg * 1e-3) / 1e9, 1),
                     "queries_rejected_by_certificate_last_step": rechecked,
                     "launch": launch},
        # (log-mel: the event pair spans k_seg_stats and the log-mel kernel; `frac` is FLOPs executed against the pipe that executes
        # them -- the fp32 vector ALU for the FFT kernel, the f16 matrix pipe otherwise --, `frac_hbm` the algorithmic bytes against HBM)
        "kernels": {lm_name: kroof(lm_avg, lm_flops, lm_bytes, lm_bound, lm_peak),
                    "k_proj_pool2": kroof(pp_avg, pp_flops, pp_bytes, "mfma", PEAK_MFMA_F16_TFLOPS)},
        "frames_transformed_per_step": n_transforms,
        "kernels_measured": "HIP events over %d steps right after the timed region (the timed region carries the scan's events only)" % min(args.steps, 20),
        "logmel_kind": logmel_kind,
        "kernels_ms": {"k_logmel": round(lm_avg, 4), "k_proj_pool": round(pp_avg, 4), "scan": round(knn_avg, 4)},
    }
    if shard_ms:     # this rank's share of a step outside the scan: collectives (query all-gather, bound all-reduce, list exchange) and
        # the re-rank half of the search (float64 re-rank + exact kernel)
        cm, rm = float(np.mean([a for a, _ in shard_ms])), float(np.mean([b for _, b in shard_ms]))
        out["sharded"] = {"collective_ms": round(max_over_ranks(cm), 4), "rerank_ms": round(max_over_ranks(rm), 4),
                          "rank0_collective_ms": round(cm, 4), "rank0_rerank_ms": round(rm, 4), "bound_exchange": bool(bounded),
                          "candidates_rescored_per_query_rank0": round(launch["certificate"]["candidates_rescored"] / max(1, Q), 1)}
    if scan_unstructured:
        out["scan_unstructured_ms"] = {k_: v_["scan_ms"] for k_, v_ in scan_unstructured.items()}
        out["scan_unstructured"] = scan_unstructured
    if sustained:
        out["sustained"] = sustained
    if pcie:
        out["pcie_inclusive"] = pcie
    if pcie_pcm16:
        out["pcie_inclusive_pcm16"] = pcie_pcm16

    # ---- CPU side (rank 0, one GPU): the float64 oracle CHECKS the GPU result; the float32 torch-CPU port is TIMED ------
    if rank == 0 and world == 1 and (args.cpu_sample > 0 or args.cpu_baseline_clips > 0):
        from oracle import radad_oracle as O
        from oracle import cpu_baseline as CB
        w0, o0 = batches[0][0], batches[0][3]
        db_h = rows.cpu().numpy()
        # threads actually used: the CPUs this process may run on, at most 16 (a one-GPU box is a 16-CPU share of its host;
        # more threads than that share only fight each other: 256 threads took 10x longer than 16) -- RADAD_CPU_THREADS overrides
        cores = int(os.environ.get("RADAD_CPU_THREADS", min(len(os.sched_getaffinity(0)), 16)))
        if args.cpu_baseline_clips > 0:
            nb = min(args.cpu_baseline_clips, B)
            wav_b = [w0[o0[b]:o0[b + 1]].cpu().numpy() for b in range(nb)]
            mel = O.mel_filter_bank().astype(np.float32)
            CB.embed_clips(wav_b[:8], fe.segment_length, fe.hop_length, fe.proj_w, fe.proj_b, mel, (1,), "max", threads=cores)   # warm-up
            t0 = time.perf_counter()
            e_cpu = CB.embed_clips(wav_b, fe.segment_length, fe.hop_length, fe.proj_w, fe.proj_b, mel, (1,), "max", threads=cores)
            t_e = time.perf_counter() - t0
            d_cpu, i_cpu = CB.knn_cosine(db_h, e_cpu, TOP_K, threads=cores)
            t_all = time.perf_counter() - t0
            agree = float(np.mean(i_cpu[:, 0] == I[:nb, 0].cpu().numpy()))
            out["cpu_baseline"] = {"value": round(nb / t_all, 2), "unit": "clips/s", "cores": cores,
                                   "affinity_cpus": len(os.sched_getaffinity(0)), "kind": "port",
                                   "cpu_model": CB.cpu_model(),
                                   "sample": f"{nb} of the {B} clips against the full {n_total} x {DIM} store: float32 torch-CPU "
                                             f"(FFT + BLAS, {cores} threads) embed {t_e:.2f} s + normalise/GEMM/top-k {t_all - t_e:.2f} s",
                                   "top1_agrees_with_gpu": agree}
        if args.cpu_sample > 0:
            ns = min(args.cpu_sample, B)
            wav_h = [w0[o0[b]:o0[b + 1]].cpu().numpy() for b in range(ns)]
            emb_ref = O.embed_clips(wav_h, fe.segment_length, fe.hop_length, fe.proj_w, fe.proj_b, (1,), "max")
            emb_gpu = emb[:ns].float().cpu().numpy()
            emb_err = float(np.abs(emb_gpu - emb_ref).max())
            od, oi = O.knn(db_h, emb_ref, TOP_K, "COSINE", chunk=65536)
            I_h, D_h = I[:ns].cpu().numpy(), D[:ns].cpu().numpy()
            recall = float(np.mean([len(set(a) & set(b)) / TOP_K for a, b in zip(I_h, oi)]))
            # retrieve parity proper: the reference normalises in float32 BEFORE the index sees the vectors
            # (vector_database.py:103-104,118,166), so the search is judged on the rows and queries as stored --
            # float64 inner products of exactly those float32 vectors, (distance, id) order.
            eq = emb[:ns].float().contiguous()
            qn = torch.empty_like(eq)
            _lib.check(lib.radad_rownorm(eq.data_ptr(), qn.data_ptr(), ns, DIM, local_rank, _lib.stream_ptr(dev)))
            stored = np.empty((hi - lo, DIM), np.float32)             # the rows exactly as the store holds them (decoded)
            for r0 in range(0, hi - lo, 131072):
                ids = torch.arange(lo + r0, min(hi, lo + r0 + 131072), device=dev)
                stored[r0:r0 + len(ids)] = vdb.index.reconstruct_batch(ids).cpu().numpy()
            sd, si = O.knn(stored, qn.cpu().numpy(), TOP_K, "IP", chunk=65536)
            del stored
            tol = 1e-4 if args.embed_dtype == "f32" else 2e-2        # bf16 embeddings carry 8 significant bits
            out["parity_on_sample"] = {"checker": f"float64 numpy oracle on {ns} clips x the full store",
                                       "ids_bit_exact": bool(np.array_equal(I_h, si)),
                                       "max_abs_dist_err": float(np.abs(D_h - sd).max()), "max_abs_embed_err": emb_err,
                                       "embed_within_tolerance": bool(emb_err < tol),
                                       "max_abs_rownorm_err": float(np.abs(qn.cpu().numpy().astype(np.float64) -
                                                                           O.maybe_normalize(eq.cpu().numpy(), True)).max()),
                                       "recall_at_k_vs_float64_cosine_of_oracle_embeddings": recall}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
