#!/bin/bash
# scan-kernel timing ablations (run through gpurun): the experimental build honours RADAD_DEBUG_KNN bits
# 1 skip epilogue, 2 skip DMA issue, 4 skip MFMAs, 8 skip barrier, 16 skip LDS reads, 32 epilogue never enters the push path,
# 64 drain does nothing.  Results are wrong when set.
export RADAD_HIP_LIB=$GRAFT_REPO_ROOT/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
for d in ${RADAD_ABLATE_SET:-0 1 2 3 4 8 16 18 19 20 22 27 32 64 96}; do
  echo -n "debug $d: "
  RADAD_DEBUG_KNN=$d python tools/exp_scan.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['scan_ms'], d['scan_TFLOPs'])"
done
