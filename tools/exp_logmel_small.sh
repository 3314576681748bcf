#!/bin/bash
# k_logmel_h ablations on a grid of ONE workgroup per CU (126 segments = 252 workgroups) and of two (255 segments), experimental build
export RADAD_HIP_LIB=$GRAFT_REPO_ROOT/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
for c in 42 85; do for d in 0 1 2 4 6 7; do
  echo -n "clips $c debug $d: "
  RADAD_DEBUG_LOGMEL=$d python bench.py --clips $c --steps 20 --warmup 3 --cpu-sample 0 --cpu-baseline-clips 0 --sustain 0 --pcie 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['kernels_ms']['k_logmel'])"
done; done
