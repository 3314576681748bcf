#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench (run through gpurun): tools/profile_stats.sh <tag>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/stats_$1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --cpu-baseline-clips 0 --sustain 0 --pcie 0 > $OUT.log 2>&1 || exit 1
ls $OUT
