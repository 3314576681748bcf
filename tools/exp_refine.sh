#!/bin/bash
# k_merge_refine timing ablations (run through gpurun; experimental build; results are wrong when a bit is set):
# RADAD_DEBUG_KNN 128 no statistics atomics, 256 no float64 re-score, 512 no ranking/output.  Kernel time from rocprofv3 --stats.
export RADAD_HIP_LIB=$GRAFT_REPO_ROOT/radad_retrievalaugmenteddeepfakeaudiodetection_amd/libradad_hip_exp.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in ${RADAD_ABLATE_SET:-0 128 256 512 896}; do
  export RADAD_DEBUG_KNN=$d
  out=gpurun_out/refine_$d
  rm -rf $out; mkdir -p $out
  timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --cpu-baseline-clips 0 --sustain 0 --pcie 0 > $out.log 2>&1
  echo -n "debug $d: "; grep "k_merge_refine" $out/p_kernel_stats.csv | cut -d, -f2-4
done
