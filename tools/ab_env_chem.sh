# A/B of an environment switch, printing the fused pass's own time: bash tools/ab_env_chem.sh ASORA_REACH_MASK "0 1 0 1 0 1" 16 32
V=$1; VALS=$2; shift 2
for X in $VALS; do for RR in "$@"; do
  env $V=$X timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-sources 0 --evolving-state 0 --R $RR > gpurun_out/abc.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/abc.json')); print('$V=$X R=$RR', 'fused pass ms', round(d['kernels_ms_per_step']['chemistry'],4), 'raytrace ms', round(d['kernels_ms_per_step']['raytrace'],4), 'step ms', round(d['ms_per_step'],4))"
done; done
