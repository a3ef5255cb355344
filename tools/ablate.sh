for A in ${ABLATE_SET:-0 1 2}; do
  ASORA_ABLATE=$A timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sources 0 $BENCH_ARGS 2>/dev/null > gpurun_out/abl_$A.json
  python -c "
import json;d=json.load(open('gpurun_out/abl_$A.json'));print('ablate $A', d['ms_per_step'], d['kernels_ms_per_step']['raytrace'])"
done
