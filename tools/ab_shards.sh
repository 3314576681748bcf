for i in 1 2; do for L in libradad_hip.so libradad_hip_old.so; do export RADAD_HIP_LIB=$PWD/radad_retrievalaugmenteddeepfakeaudiodetection_amd/$L; echo $L; for w in 2 4 8; do python tools/rehearse_rank.py --world $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['world'], d['global_bound'], d['scan_ms'], d['rerank_ms'], d['search_ms'], d['launch'].get('scan_launches'))"; done; python bench.py --db-rows 100000 --cpu-baseline-clips 0 --pcie 0 --sustain 0 --cpu-sample 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('config2', round(d['value']/1e3,1), d['ms_per_step'], d['kernels_ms'])"; done; done
