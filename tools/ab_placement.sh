# The default bench job with and without the placement probe of device_init (ASORA_PLACEMENT_CANDIDATES = 1: first allocation taken),
# alternating on ONE box: every process places its grids anew.   bash tools/ab_placement.sh [rounds]
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
mkdir -p gpurun_out
for ROUND in $(seq 1 ${1:-5}); do
for C in 1 8; do
    ASORA_PLACEMENT_CANDIDATES=$C timeout -k 10 300 python bench.py --steps 20 --warmup 5 --repeats 3 --cpu-sources 0 --evolving-state 0 > gpurun_out/abp.json 2>/dev/null || { echo "candidates=$C FAILED"; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/abp.json")); k=d["kernels_ms_per_step"]; g=d["config"]["grid_placement"]
print("round $ROUND candidates<=$C: ms/step %.4f  raytrace %.4f  fused pass %.4f   tried %d, probe %.3f ms (slowest %.3f)" % (d["ms_per_step"], k["raytrace"], k["chemistry"], g["candidates"], g["chosen_probe_ms"], g["slowest_probe_ms"]))
PY
done
done
