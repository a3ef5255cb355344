# Overhead of the pipelined raytrace + all-reduce schedule on ONE GPU (collective forced, world size 1):
# what chunking the sources and folding slab by slab costs when there is nothing to overlap with.
export PYC2RAY_AMD_FORCE_COLLECTIVE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
for OV in 0 1; do
  MASTER_PORT=$((29533+OV)) timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-sources 0 --overlap $OV > gpurun_out/ov$OV.json 2> gpurun_out/ov$OV.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ov$OV.json")); print("overlap $OV", d["value"], d["ms_per_step"], d["kernels_ms_per_step"], d["config"]["parallelism"])
PY
done
