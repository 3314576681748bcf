#!/bin/bash
# A/B on ONE box: the scan in one launch (floors raised inside it) against round 4's launch per phase, three alternations
set -e
cd "$(dirname "$0")/.."
out=gpurun_out/r5_ab.txt
: > $out
for i in 1 2 3; do
  for lf in ${LFS:-1 0}; do
    for rows in ${ROWS:-1000000 100000}; do
      python bench.py --steps 60 --warmup 10 --db-rows $rows --live-floor $lf --sustain 0 --pcie 0 --unstructured 0 --cpu-sample 0 --cpu-baseline-clips 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('live_floor $lf rows $rows value', d['value'], 'ms', d['ms_per_step'], 'scan_ms', r['scan_ms_per_step'], 'launches', r['launches_per_step'], 'frac', r['frac'], 'rej', r['launch']['certificate']['rejected'], 'resc', r['launch']['certificate']['candidates_rescored'])" >> $out
    done
  done
done
cat $out
