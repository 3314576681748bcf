#!/bin/bash
cd "$(dirname "$0")/.."
export RADAD_HIP_LIB=$PWD/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
for rep in 1 2; do for d in 0 32 1; do RADAD_DEBUG_KNN=$d python tools/exp_scan.py --reps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('RADAD_DEBUG_KNN=$d scan_ms', d['scan_ms'], 'TF', d['scan_TFLOPs'], d['launch']['certificate'])"; done; done | tee gpurun_out/r5_ablate2.txt
