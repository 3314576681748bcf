#!/bin/bash
# A/B of two builds of the library on ONE box (run through gpurun): libradad_hip.so against libradad_hip_old.so, alternating
for i in 1 2 3; do for L in libradad_hip.so libradad_hip_old.so; do RADAD_HIP_LIB=$PWD/radad_retrievalaugmenteddeepfakeaudiodetection_amd/$L python bench.py --steps 20 --warmup 3 --cpu-sample 0 --cpu-baseline-clips 0 --sustain 0 --pcie 0 ${AB_ARGS} > gpurun_out/b_$L.$i.log 2>&1; tail -1 gpurun_out/b_$L.$i.log | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', round(d['value']/1e3,1), d['kernels_ms'], {k:v['scan_ms'] for k,v in (d.get('scan_unstructured') or {}).items()})"; done; done
