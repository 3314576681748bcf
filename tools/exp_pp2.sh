#!/bin/bash
# k_proj_pool2 timing experiments (run through gpurun) on the experimental build (make -C .../csrc exp):
#   RADAD_DEBUG_LOGMEL bits seen by k_proj_pool2: 16 no MFMAs, 32 no pooling, 64 no conversion of the next pass (results wrong when set)
export RADAD_HIP_LIB=$GRAFT_REPO_ROOT/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
for d in ${RADAD_ABLATE_SET:-0 16 32 64 48 80 96 112}; do echo -n "debug $d: "; RADAD_DEBUG_LOGMEL=$d python tools/exp_embed.py 2>/dev/null | tail -1; done
