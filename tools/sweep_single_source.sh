# BASELINE configs[1]: one source, 128^3, r_RT = 64: raytrace time per launch for every decomposition / workgroup size
for M in 0 1 2 3; do for T in 0 256 512 1024; do
  if [ $M = 0 ] && [ $T != 0 ]; then continue; fi
  if [ $M != 0 ] && [ $T = 0 ]; then continue; fi
  timeout -k 10 300 python bench.py --N 128 --nsrc 1 --R 64 --steps 20 --warmup 3 --cpu-sources 0 --sectors $M --block-threads $T 2>/dev/null > gpurun_out/ss_${M}_$T.json
  python - <<PY
import json
d=json.load(open("gpurun_out/ss_${M}_$T.json")); print("mode=$M threads=$T raytrace_ms=%.4f step_ms=%.4f" % (d["kernels_ms_per_step"]["raytrace"], d["ms_per_step"]))
PY
done; done
