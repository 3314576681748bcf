#!/bin/bash
# configs 4 and 5 at full size: one scan launch (floors raised inside) against one launch per phase, on ONE box
cd "$(dirname "$0")/.."
for c in 4 5; do for lf in 1 0 1 0; do
  python tools/bench_fullstore.py --config $c --live-floor $lf --steps 4 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('config $c live_floor $lf value', d['value'], 'ms', d['ms_per_step'], 'search_ms', d['search_ms'], 'scan_ms', r['scan_ms_per_search'], 'launches', r['launches_per_search'], 'phases', r['launch']['scan_phases'], 'frac', r['frac'], 'rej', r['launch']['certificate']['rejected'], 'tuning', d['tuning'], 'found', d['config']['planted_neighbours_found'], d['config']['planted_row_is_top1'])"
done; done | tee gpurun_out/r5_ab_full.txt
