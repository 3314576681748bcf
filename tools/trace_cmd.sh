#!/bin/bash
# rocprofv3 kernel-trace + stats of any python command (run through gpurun): tools/trace_cmd.sh <tag> <script> [args...]
# prints the per-kernel table (name, calls, average ns, share)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/trace_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 "$@" > $OUT.log 2>&1 || { tail -n 20 $OUT.log; exit 1; }
tail -n 3 $OUT.log
f=$(find $OUT -name '*kernel_stats.csv' | head -n 1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.2f} us {float(r["Percentage"]):6.2f} %')
PY
