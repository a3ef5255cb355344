# workgroup-size / decomposition sweep of the raytrace kernel per radius
for RR in ${RADII:-16 32 64}; do for M in ${MODES:-1 2}; do for T in ${THREADS:-64 128 256 512}; do
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-sources 0 --R $RR --block-threads $T --sectors $M 2>/dev/null > gpurun_out/sw_${RR}_${M}_$T.json
  python -c "
import json;d=json.load(open('gpurun_out/sw_${RR}_${M}_$T.json'));print('R=$RR mode=$M threads=$T raytrace_ms=%.3f evals=%d'%(d['kernels_ms_per_step']['raytrace'], d['config']['column_density_evaluations_per_step_rank0']))"
done; done; done
