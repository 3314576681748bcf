#!/bin/bash
# small-batch search visit: kNN tests, then the online-search bench lines (reference shape, 1 M x 512) -- tools/r4_small.sh <tag> ["<pytest selection>"]
tag=$1; sel=${2:-tests/test_gpu_knn.py tests/test_gpu_reference_shapes.py tests/test_gpu_ivf.py}
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest $sel -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -n 8 gpurun_out/${tag}_tests.log
if [ $rc -ne 0 ]; then echo "tests failed (rc $rc): no further GPU step"; exit $rc; fi
for cfg in "--db-rows 25423 --dim 5376 --metric l2 --k 15" "--db-rows 25423 --dim 3584 --metric cosine --k 15" "--k 15" "--predict-queries 16"; do
  timeout -k 10 300 python bench.py --mode predict $cfg --steps 200 --warmup 10 > gpurun_out/${tag}_pred.json 2> gpurun_out/${tag}_pred.err || { tail -n 5 gpurun_out/${tag}_pred.err; exit 1; }
  python - "$cfg" gpurun_out/${tag}_pred.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "->", d["value"], d["unit"], "ms/search", d["ms_per_step"], "scan", d["roofline"].get("kernel_ms"), "planted", d.get("planted_neighbours_found"))
PY
done
