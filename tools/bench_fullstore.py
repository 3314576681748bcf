#!/usr/bin/env python3
"""BASELINE configs 4 and 5 at full store size on ONE MI355X (the replicated layout of DESIGN section 6 holds these stores whole):

  --config 4 : 10 240 clips (4 s @16 kHz) -> F = 512 fp32 embeddings -> cosine top-10 against 10 M x 512 fp32 (+ f16 plane)
  --config 5 : 10 240 clips -> F = 256 bf16 embeddings -> cosine top-10 against 50 M x 256 fp16, fp32 accumulate

Prints one JSON line in bench.py's format (value = clips/s for embed + search of the whole batch; roofline = the scan).
Parity of these shapes is tests/test_gpu_fullstore.py's business; here only the planted neighbours are checked."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, choices=[4, 5], required=True)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--clips", type=int, default=10240)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--plant-every", type=int, default=8)
    ap.add_argument("--live-floor", type=int, default=-1, help="-1 = the library's default, 1 / 0 = one scan launch with the floors raised inside it / one launch per phase")
    args = ap.parse_args()
    import numpy as np
    import torch
    import radad_retrievalaugmenteddeepfakeaudiodetection_amd as R
    from radad_retrievalaugmenteddeepfakeaudiodetection_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    c5 = args.config == 5
    dim = 256 if c5 else 512
    n = args.rows or (50_000_000 if c5 else 10_000_000)
    B, k = args.clips, args.k
    cfg = R.Config()
    cfg.update(device=dev, tpp_levels=[1], tpp_pooling_type="max", feature_dim=dim, vector_db_index_type="IP", use_float16=c5)
    fe = R.MelProjectionFeatureExtractor(cfg)
    wave = torch.empty(B * 64000, device=dev)
    _lib.check(lib.radad_synth_audio(wave.data_ptr(), 0, B, 64000, 1234, 0, _lib.stream_ptr(dev)))
    offs = np.arange(B + 1, dtype=np.int64) * 64000
    emb_dtype = torch.bfloat16 if c5 else torch.float32
    emb0 = fe.embed_clips(wave, offs).float()
    scale = emb0.norm(dim=1, keepdim=True) / dim ** 0.5
    # near-duplicates are planted for every 8th query only: synthetic clips embed within cos 0.99 of each other, so EVERY planted row
    # neighbours EVERY query -- 10 240 of them put ~1300 rows within 2 eps of each query's 10th best (the default bench plants 2048 for
    # 1024 queries: ~150).  --plant-every 1 measures that store: the handle widens its buffers (cap_boost) inside the first search.
    pq = torch.arange(0, B, args.plant_every, device=dev)
    planted = (pq * (n // B - 3) + 29) % n
    idx = R.HipFlatIndex(dim, _lib.METRIC_COSINE, 0, store_f16=c5, live_floor=(None if args.live_floor < 0 else args.live_floor))
    t_add = time.perf_counter()
    step = 1 << 20
    for r0 in range(0, n, step):
        m = min(step, n - r0)
        rows = torch.empty((m, dim), device=dev)
        _lib.check(lib.radad_synth_rows(rows.data_ptr(), r0, m, dim, 4321, 0, _lib.stream_ptr(dev)))
        sel = (planted >= r0) & (planted < r0 + m)
        ns = int(sel.sum())
        if ns:
            rows[planted[sel] - r0] = emb0[pq[sel]] + 0.05 * scale[pq[sel]] * rows[:ns]
        idx.add_device(rows)
    torch.cuda.synchronize()
    t_add = time.perf_counter() - t_add
    del rows

    def step_fn():
        e = fe.embed_clips(wave, offs, out_dtype=emb_dtype)
        return idx.search_device(e, k)
    for _ in range(args.warmup):
        step_fn()
    torch.cuda.synchronize()
    idx.profile(True, every=1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        D, I = step_fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scan = idx.profile_read()
    launch = idx.last_launch()
    nl = max(1, launch["scan_launches"])
    scan_ms = float(np.mean(scan)) * nl if scan else float("nan")
    # search alone (embeddings resident)
    e = fe.embed_clips(wave, offs, out_dtype=emb_dtype)
    idx.profile(False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        idx.search_device(e, k)
    torch.cuda.synchronize()
    search_ms = 1e3 * (time.perf_counter() - t1) / args.steps
    flops = 2.0 * B * n * dim
    plane_bytes = 2.0 * n * dim
    qtiles = (B + 255) // 256
    achieved = flops / (scan_ms * 1e-3) / 1e12
    out = {"metric": f"clips/sec (segment+embed+retrieve) @{n}x{dim} DB (BASELINE config {args.config} at full store size, one GPU)",
           "value": round(B * args.steps / dt, 1), "unit": "clips/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": ("bf16 embeddings / f16 store" if c5 else "f32 embeddings / f32 store + f16 plane") +
                    " (certified f16 MFMA scan, f32 accumulate, float64 re-rank)",
           "data": "synthetic",
           "config": {"workload": f"{B} clips x 4 s @16 kHz, F={dim}, levels=[1], cosine top-{k}, {n} x {dim} "
                                  f"{'f16' if c5 else 'f32'} store on ONE handle", "db_rows": n, "dim": dim, "k": k,
                      "planted_for_every_nth_query": args.plant_every, "live_floor": args.live_floor,
                      # (synthetic clips come in 181 periods: clips of one period embed within cos 0.9999 of each other, so the top hit of a
                      # query may be the planted copy of a same-period clip -- with bf16 embeddings it often is)
                      "planted_row_is_top1": round(float((I[pq, 0] == planted).float().mean().item()), 4),
                      "planted_neighbours_found": bool((I[pq] == planted[:, None]).any(dim=1).all().item())},
           "roofline": {"kernel": "k_knn_hi", "bound": "mfma", "achieved": round(achieved, 1), "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": round(achieved / 2500.0, 4), "traffic": None, "scan_ms_per_search": round(scan_ms, 3),
                        "launches_per_search": nl, "flops_per_search": flops,
                        "plane_bytes_streamed_per_query_tile_pass": plane_bytes, "query_tiles": qtiles,
                        "launch": launch},
           "search_ms": round(search_ms, 3), "embed_ms": round(1e3 * dt / args.steps - search_ms, 3),
           "store_build_s": round(t_add, 2), "plane": idx.plane_info(), "tuning": idx.tuning_info(),
           "hbm_allocated_GB": round(torch.cuda.mem_get_info(0)[1] / 1e9 - torch.cuda.mem_get_info(0)[0] / 1e9, 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
