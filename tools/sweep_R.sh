# raytrace kernel time per launch for a list of radii (default launch shape): bash tools/sweep_R.sh 16 20 24 32
for RR in "$@"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sources 0 --R $RR > gpurun_out/sweepR_$RR.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/sweepR_$RR.json")); print("R=$RR", "raytrace ms", round(d["kernels_ms_per_step"]["raytrace"],4), "step ms", round(d["ms_per_step"],4), "ns/src/cell", round(d["raytrace_ns_per_source_per_insphere_cell"],5))
PY
done
