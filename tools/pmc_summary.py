#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean of each counter over the dispatches of our kernels.
usage: pmc_summary.py <dir with *_counter_collection.csv> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

KEEP = ("k_knn_hi_sample", "k_knn_hi_smallq", "k_knn_hi", "k_knn_f32", "k_exact_scan", "k_exact_merge", "k_hi_rows", "k_build_plan", "k_seg_stats", "k_logmel_fft_clip", "k_proj_pool2", "k_logmel_h_clip", "k_logmel", "k_proj_pool", "k_merge_refine", "k_rows_prepare", "k_split_rows", "k_thr_from_parts", "k_floor_from_sample", "k_kth_floor")
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                short = next((k for k in KEEP if k in name), None)
                if short is None:
                    continue
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
                for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
                    if extra in row and row[extra] != "":
                        acc[short]["_" + extra] = [float(row[extra])]
        print(f"# {f}")
        for k in KEEP:
            if k in acc:
                print(k)
                for c, v in sorted(acc[k].items()):
                    print(f"  {c:36s} mean {sum(v) / len(v):.6g}  (n={len(v)})")
