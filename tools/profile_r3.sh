#!/bin/bash
# Round-3 profile set (run through gpurun): kernel stats, PMC passes, and the bench records copied to profiles/ afterwards.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=$1
echo stats; timeout -k 10 300 bash tools/profile_stats.sh ${T} > /dev/null 2>&1
echo pmc; timeout -k 10 600 bash tools/profile_r1.sh ${T} > gpurun_out/pmc_${T}.out 2>&1
echo benches
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
python bench.py --workload ragged --pcie 0 > gpurun_out/${T}_bench_ragged.json 2> /dev/null
python bench.py --db-rows 100000 --cpu-baseline-clips 0 --pcie 0 --sustain 0 > gpurun_out/${T}_bench_config2.json 2> /dev/null
python bench.py --store-dtype f16 --cpu-baseline-clips 0 --pcie 0 --sustain 0 > gpurun_out/${T}_bench_f16.json 2> /dev/null
python bench.py --store-dtype f16 --embed-dtype bf16 --cpu-baseline-clips 0 --pcie 0 --sustain 0 > gpurun_out/${T}_bench_bf16_f16.json 2> /dev/null
python bench.py --scan f32 --cpu-sample 0 --cpu-baseline-clips 0 --pcie 0 --sustain 0 --unstructured 0 > gpurun_out/${T}_bench_scan_f32.json 2> /dev/null
python bench.py --mode predict --steps 200 --warmup 10 > gpurun_out/${T}_predict_1.json 2> /dev/null
python bench.py --mode predict --predict-queries 16 --steps 200 --warmup 10 > gpurun_out/${T}_predict_16.json 2> /dev/null
for d in 5376 3584; do for m in l2 cosine; do
python bench.py --mode predict --dim $d --db-rows 25423 --metric $m --k 15 --steps 200 --warmup 10 > gpurun_out/${T}_predict_ref_${d}_${m}.json 2> /dev/null
done; done
python tools/exp_scan.py --rows 25423 --dim 5376 --nq 256 --k 15 --metric l2 --reps 20 > gpurun_out/${T}_refshape_scan.txt 2> /dev/null
python tools/exp_scan.py --rows 25423 --dim 5376 --nq 256 --k 15 --metric cosine --reps 20 >> gpurun_out/${T}_refshape_scan.txt 2> /dev/null
python tools/exp_scan.py --rows 25423 --dim 3584 --nq 256 --k 15 --metric l2 --reps 20 >> gpurun_out/${T}_refshape_scan.txt 2> /dev/null
for m in cosine ip l2; do python tools/exp_scan.py --metric $m --reps 20 2> /dev/null; done > gpurun_out/${T}_scan_unstructured.txt
for w in 2 4 8; do python tools/rehearse_rank.py --world $w 2> /dev/null; done > gpurun_out/${T}_rehearse.txt
python tools/bench_ivf.py > gpurun_out/${T}_ivf.json 2> /dev/null
for f in default ragged config2 f16 bf16_f16 scan_f32; do python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["kernels_ms"], d.get("scan_unstructured_ms"), d.get("sustained",{}).get("value"), d.get("pcie_inclusive",{}).get("value"), d.get("cpu_baseline",{}).get("value"), d.get("parity_on_sample",{}).get("ids_bit_exact"), d["roofline"]["launch"].get("certificate"))
PY
done
for f in predict_1 predict_16 predict_ref_5376_l2 predict_ref_5376_cosine predict_ref_3584_l2 predict_ref_3584_cosine; do python -c "
import json; d=json.load(open('gpurun_out/${T}_$f.json')); print('$f', d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms'], d['roofline']['launch']['scan_kind'])"; done
cat gpurun_out/${T}_refshape_scan.txt | cut -c1-200
