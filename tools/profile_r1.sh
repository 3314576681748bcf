#!/bin/bash
# rocprofv3 passes for the round-1 profile (run on the GPU box through gpurun; writes under gpurun_out/).
# Counters are collected in their own runs with --kernel-trace only (no sys/hip/hsa tracing beside --pmc).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
ARGS="bench.py --steps 3 --warmup 1 --cpu-sample 0 --cpu-baseline-clips 0 --sustain 0 --pcie 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq1 -o p -- python3 $ARGS > $OUT.sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -o p -- python3 $ARGS > $OUT.sq2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -o p -- python3 $ARGS > $OUT.fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o p -- python3 $ARGS > $OUT.write.log 2>&1 || exit 1
python3 tools/pmc_summary.py $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write > $OUT.summary.txt 2>&1
tail -80 $OUT.summary.txt
