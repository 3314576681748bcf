#!/bin/bash
# rocprofv3 PMC passes for an arbitrary python tool: tools/pmc_cmd.sh <tag> <script.py> [args...]   (run through gpurun)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA --output-format csv -d $OUT/sq1 -o p -- python3 "$@" > $OUT.sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -o p -- python3 "$@" > $OUT.sq2.log 2>&1 || exit 1
# (FETCH_SIZE + TCC_HIT + TCC_MISS + TCP_TCC_READ_REQ + GRBM in ONE pass is more than the hardware collects at once: rocprofv3 aborts
# with "error code 38: Request exceeds the capabilities of the hardware to collect" -- gpurun_out/pmc_r4a_fft.fetch.log; two passes,
# as tools/profile_r4.sh splits them)
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -o p -- python3 "$@" > $OUT.fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o p -- python3 "$@" > $OUT.write.log 2>&1 || exit 1
python3 tools/pmc_summary.py $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write > $OUT.summary.txt 2>&1
