#!/bin/bash
# scan-kernel ablations on the BENCHMARK's data (run through gpurun; experimental build; results are wrong when a bit is set):
# RADAD_DEBUG_KNN bits as in tools/exp_ablate.sh
export RADAD_HIP_LIB=$GRAFT_REPO_ROOT/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
for d in ${RADAD_ABLATE_SET:-0 1 32 64 96}; do
  echo -n "debug $d: "
  RADAD_DEBUG_KNN=$d python bench.py --steps 10 --warmup 3 --cpu-sample 0 --sustain 0 --pcie 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['kernels_ms'], d['roofline']['launch']['certificate'])"
done
