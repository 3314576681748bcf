#!/bin/bash
# Round-5 profile set (run through gpurun, two calls): tools/profile_r5.sh <tag> pmc | bench
#   pmc   : rocprofv3 kernel-trace + stats of the default bench, then the four PMC passes (each alone, --kernel-trace only), with a
#           heartbeat on stdout while a pass runs (a silent command is taken for hung after 7 minutes)
#   bench : the bench records (default, ragged, config 2, fp16 / bf16, fp32 scan, online search lines, predict() end to end, IVF, rehearsal)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=$1; what=$2
mkdir -p gpurun_out
beat() {   # beat <log> <cmd...>: run in the background, a dot every 30 s
    local log=$1; shift
    "$@" > $log 2>&1 &
    local pid=$!
    while kill -0 $pid 2>/dev/null; do sleep 30; echo -n "."; done
    wait $pid
}
if [ "$what" = pmc ]; then
    ARGS="bench.py --steps 3 --warmup 1 --cpu-sample 0 --cpu-baseline-clips 0 --sustain 0 --pcie 0 --unstructured 0"
    OUT=gpurun_out/pmc_$T
    echo stats; beat gpurun_out/stats_$T.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$T -o p -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --cpu-baseline-clips 0 --sustain 0 --pcie 0 || exit 1
    echo sq1; beat $OUT.sq1.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq1 -o p -- python3 $ARGS || exit 1
    echo sq2; beat $OUT.sq2.log rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS || exit 1
    echo fetch; beat $OUT.fetch.log rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -o p -- python3 $ARGS || exit 1
    echo write; beat $OUT.write.log rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o p -- python3 $ARGS || exit 1
    python3 tools/pmc_summary.py $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write > $OUT.summary.txt 2>&1
    tail -n 60 $OUT.summary.txt
    python3 tools/trace_seq.py gpurun_out/stats_$T k_logmel_fft_clip 8 2 14 > gpurun_out/${T}_step_trace.txt 2>&1
    exit 0
fi
echo benches
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo default
python bench.py --workload ragged --pcie 0 > gpurun_out/${T}_bench_ragged.json 2> /dev/null; echo ragged
python bench.py --db-rows 100000 --cpu-baseline-clips 0 --pcie 0 --sustain 0 > gpurun_out/${T}_bench_config2.json 2> /dev/null; echo config2
python bench.py --store-dtype f16 --cpu-baseline-clips 0 --pcie 0 --sustain 0 > gpurun_out/${T}_bench_f16.json 2> /dev/null
python bench.py --store-dtype f16 --embed-dtype bf16 --cpu-baseline-clips 0 --pcie 0 --sustain 0 > gpurun_out/${T}_bench_bf16_f16.json 2> /dev/null; echo f16
python bench.py --scan f32 --cpu-sample 0 --cpu-baseline-clips 0 --pcie 0 --sustain 0 --unstructured 0 > gpurun_out/${T}_bench_scan_f32.json 2> /dev/null
python bench.py --logmel gemm --cpu-sample 0 --cpu-baseline-clips 0 --pcie 0 --sustain 0 --unstructured 0 > gpurun_out/${T}_bench_logmel_gemm.json 2> /dev/null; echo ab
python bench.py --mode predict --steps 200 --warmup 10 > gpurun_out/${T}_predict_1.json 2> /dev/null
python bench.py --mode predict --k 15 --steps 200 --warmup 10 > gpurun_out/${T}_predict_1_k15.json 2> /dev/null
python bench.py --mode predict --predict-queries 16 --steps 200 --warmup 10 > gpurun_out/${T}_predict_16.json 2> /dev/null
for d in 5376 3584; do for m in l2 cosine; do
python bench.py --mode predict --dim $d --db-rows 25423 --metric $m --k 15 --steps 200 --warmup 10 > gpurun_out/${T}_predict_ref_${d}_${m}.json 2> /dev/null
done; done
python bench.py --mode predict --end-to-end --steps 200 --warmup 10 > gpurun_out/${T}_predict_e2e.json 2> /dev/null; echo predict
python tools/exp_scan.py --rows 25423 --dim 5376 --nq 256 --k 15 --metric l2 --reps 20 > gpurun_out/${T}_refshape_scan.txt 2> /dev/null
python tools/exp_scan.py --rows 25423 --dim 5376 --nq 256 --k 15 --metric cosine --reps 20 >> gpurun_out/${T}_refshape_scan.txt 2> /dev/null
python tools/exp_scan.py --rows 25423 --dim 3584 --nq 256 --k 15 --metric l2 --reps 20 >> gpurun_out/${T}_refshape_scan.txt 2> /dev/null
for m in cosine ip l2; do python tools/exp_scan.py --metric $m --reps 20 2> /dev/null; done > gpurun_out/${T}_scan_unstructured.txt; echo scans
for w in 2 4 8; do python tools/rehearse_rank.py --world $w 2> /dev/null; done > gpurun_out/${T}_rehearse.txt
for w in 2 4 8; do python tools/rehearse_rank.py --world $w --plant all 2> /dev/null; done > gpurun_out/${T}_rehearse_plant_all.txt; echo rehearse
python tools/bench_fullstore.py --config 4 > gpurun_out/${T}_config4_full.json 2> /dev/null
python tools/bench_fullstore.py --config 5 > gpurun_out/${T}_config5_full.json 2> /dev/null; echo fullstore
python bench.py --live-floor 1 --cpu-sample 0 --cpu-baseline-clips 0 --pcie 0 --sustain 0 --unstructured 0 > gpurun_out/${T}_bench_one_launch.json 2> /dev/null
python bench.py --live-floor 1 --db-rows 100000 --cpu-sample 0 --cpu-baseline-clips 0 --pcie 0 --sustain 0 --unstructured 0 > gpurun_out/${T}_bench_config2_one_launch.json 2> /dev/null; echo ab2
python tools/bench_ivf.py > gpurun_out/${T}_ivf.json 2> /dev/null
python tools/bench_ivf.py 1,256,1024 0 > gpurun_out/${T}_ivf_f32_lists.json 2> /dev/null; echo ivf
for f in default ragged config2 f16 bf16_f16 scan_f32 logmel_gemm; do python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["kernels_ms"], d["roofline"]["frac"], d.get("scan_unstructured_ms"), d.get("sustained",{}).get("value"), d.get("pcie_inclusive",{}).get("value"), d.get("cpu_baseline",{}).get("value"), d.get("parity_on_sample",{}).get("ids_bit_exact"), d["roofline"]["launch"].get("certificate"))
PY
done
for f in predict_1 predict_1_k15 predict_16 predict_ref_5376_l2 predict_ref_5376_cosine predict_ref_3584_l2 predict_ref_3584_cosine; do python -c "
import json; d=json.load(open('gpurun_out/${T}_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel_ms'], d['roofline']['launch']['scan_kind'])"; done
python -c "
import json; d=json.load(open('gpurun_out/${T}_predict_e2e.json')); print('predict_e2e', {k: d[k] for k in ('value','unit','ms_per_step') if k in d})"
python -c "
import json
for f in ('ivf','ivf_f32_lists'):
    d=json.load(open('gpurun_out/${T}_'+f+'.json'))
    print(f, {k:(v['search_ms'], v['flat_search_ms'], v['scan'], v['rejected'], v['roofline']['frac']) for k,v in d.items() if k.startswith('nq')})"
cut -c1-200 gpurun_out/${T}_refshape_scan.txt
