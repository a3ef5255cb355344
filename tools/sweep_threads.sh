# workgroup-size sweep of the raytrace kernel per radius
for RR in ${RADII:-16 32 64}; do for T in ${THREADS:-64 128 256 512}; do
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-sources 0 --R $RR --block-threads $T 2>/dev/null > gpurun_out/sw_${RR}_$T.json
  python -c "
import json;d=json.load(open('gpurun_out/sw_${RR}_$T.json'));print('R=$RR threads=$T raytrace_ms=%.3f'%d['kernels_ms_per_step']['raytrace'])"
done; done
