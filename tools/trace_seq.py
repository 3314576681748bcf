#!/usr/bin/env python3
"""Print the kernels around the n-th launch of an anchor kernel from a rocprofv3 --kernel-trace CSV (start offset, duration, grid):
tools/trace_seq.py <dir> <anchor substring> [occurrence=10] [before=8] [after=14]"""
import csv, glob, sys
d, key = sys.argv[1], sys.argv[2]
occ_n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
before = int(sys.argv[4]) if len(sys.argv) > 4 else 8
after = int(sys.argv[5]) if len(sys.argv) > 5 else 14
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
occ = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = occ[min(occ_n, len(occ) - 1)]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[max(0, i0 - before):i0 + after]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:6.1f}"
    prev_end = max(e, prev_end or e)
    print(f'{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us {gap:12s} grid {int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])):>6d}x{r["Workgroup_Size_X"]:>4s}  {r["Kernel_Name"][:90]}')
