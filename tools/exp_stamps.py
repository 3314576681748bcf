#!/usr/bin/env python3
"""In-kernel cycle stamps of the scan kernel (experimental build, through gpurun):
RADAD_HIP_LIB=.../libradad_hip_exp.so RADAD_KNN_STAMPS=gpurun_out/stamps.bin python tools/exp_scan.py --reps 1 ; python tools/exp_stamps.py gpurun_out/stamps.bin
Tags: 1 stage start (first fragment reads issued), 2 after the A-side DMA issue, 3 after k step 0, 4 after the B-side DMA issue,
5 after k step 1, 10 own DMA landed (vmcnt(0)), 6 after the barrier, 7 after the deferred drain, 8 tile's K loop done, 9 epilogue done."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], np.uint64).reshape(-1, 4096)
names = {1: "reads0", 2: "dmaA", 3: "kstep0", 4: "dmaB", 5: "kstep1", 6: "barrier", 7: "drain", 8: "kloop_end", 9: "epilogue", 10: "landed"}
for w in range(raw.shape[0]):
    v = raw[w]
    v = v[v != 0]
    tag = (v >> np.uint64(56)).astype(int)
    t = (v & np.uint64((1 << 56) - 1)).astype(np.int64)
    dt = np.diff(t)
    print(f"wave {w}: {len(v)} stamps, total {t[-1] - t[0]} cycles")
    # mean duration of the interval ENDING at each tag
    for k in sorted(names):
        sel = tag[1:] == k
        if sel.any():
            d = dt[sel]
            print(f"  -> {names[k]:10s} n={sel.sum():5d} mean {d.mean():8.1f} median {np.median(d):8.1f} p90 {np.percentile(d, 90):8.1f} sum {d.sum():9d}")
    # arrival at the barrier (own DMA landed, tag 10) relative to the previous barrier's release (tag 6), and the wait there
    rel, wait, last = [], [], None
    for k, tt in zip(tag, t):
        if k == 6:
            if last is not None and last[0] == 10:
                wait.append(tt - last[1])
            start = tt
        elif k == 10 and "start" in dir():
            rel.append(tt - start)
        last = (k, tt)
    if rel:
        print(f"  arrival after release: mean {np.mean(rel):8.1f} median {np.median(rel):8.1f}   wait at barrier: mean {np.mean(wait):8.1f} median {np.median(wait):8.1f}")
