#!/bin/bash
# k_logmel_fft_clip timing experiments (run through gpurun) on the experimental build (make -C .../csrc exp):
#   RADAD_DEBUG_LOGMEL bits: 1 skip the transform, 2 no sample loads, 4 no mel atomics, 8 no log-mel stores, 16 no exchange stages,
#   32 no 25-point FFTs, 64 no split / power / mel stage (results wrong when set)
export RADAD_HIP_LIB=$GRAFT_REPO_ROOT/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
for d in ${RADAD_ABLATE_SET:-0 1 2 4 8 16 32 64 127}; do echo -n "debug $d: "; RADAD_DEBUG_LOGMEL=$d python tools/exp_embed.py 2>/dev/null | tail -1; done
