#!/bin/bash
# log-mel kernel timing experiments (run through gpurun) on the experimental build (make -C .../csrc exp):
#   RADAD_DEBUG_LOGMEL bits: 1 skip the DFT MFMA loop, 2 skip the segment loads of the prologue, 4 skip the mel MFMAs, 8 skip the epilogue (log10, stores) (results wrong when set)
export RADAD_HIP_LIB=$GRAFT_REPO_ROOT/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
run() { python bench.py --steps 10 --warmup 3 --cpu-sample 0 --sustain 0 --pcie 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['kernels_ms'])"; }
for d in ${RADAD_ABLATE_SET:-0 1 2 3 8 9 11}; do echo -n "debug $d: "; RADAD_DEBUG_LOGMEL=$d run; done
