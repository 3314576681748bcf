#!/bin/bash
# one GPU visit of round 4: tests given as $1 (a pytest selection), then the default bench line; logs under gpurun_out/<tag>_*
# usage: tools/r4_run.sh <tag> "<pytest args>" ["<bench args>"]
tag=$1; sel=$2; bargs=${3:---steps 20 --warmup 5}
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest $sel -x -q > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -n 15 gpurun_out/${tag}_tests.log
if [ $rc -ge 124 ]; then echo "tests were killed or crashed (rc $rc): no further GPU step"; exit $rc; fi
timeout -k 10 400 python bench.py $bargs > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
brc=$?
tail -c 3000 gpurun_out/${tag}_bench.json
tail -n 5 gpurun_out/${tag}_bench.err
exit $(( rc != 0 ? rc : brc ))
