#!/bin/bash
# in-kernel cycle stamps of the scan (experimental build): where a stage's cycles go, per wave of workgroup 0
cd "$(dirname "$0")/.."
export RADAD_HIP_LIB=$PWD/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
RADAD_KNN_STAMPS=gpurun_out/r5_stamps.bin python tools/exp_scan.py --reps 1 > gpurun_out/r5_stamps_scan.json 2>/dev/null
python tools/exp_stamps.py gpurun_out/r5_stamps.bin > gpurun_out/r5_stamps.txt 2>&1
# ablations on the same build: 1 = skip the epilogue, 2 = skip the DMA issue (bare MFMA + LDS reads), 3 = both
for d in 0 1 2 3; do RADAD_DEBUG_KNN=$d python tools/exp_scan.py --reps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('RADAD_DEBUG_KNN=$d scan_ms', d['scan_ms'], 'TF', d['scan_TFLOPs'])"; done > gpurun_out/r5_ablate.txt
cat gpurun_out/r5_ablate.txt; head -14 gpurun_out/r5_stamps.txt
