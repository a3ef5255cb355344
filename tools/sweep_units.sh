# raytrace kernel time for explicit decompositions x workgroup sizes: bash tools/sweep_units.sh "16 24 32" "0:0 6:128 6:256 7:256"
# (each item is sectors:threads, 0:0 = the library's own choice)
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
mkdir -p gpurun_out
for RR in $1; do for ST in $2; do
  S=${ST%%:*}; T=${ST##*:}
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --R $RR --sectors $S --block-threads $T --pair-sources ${PAIRS:-1} > gpurun_out/swu.json 2>/dev/null || { echo "R=$RR sectors=$S threads=$T FAILED"; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/swu.json")); print("R=$RR sectors=$S threads=$T", "raytrace ms", round(d["kernels_ms_per_step"]["raytrace"],4), "step ms", round(d["ms_per_step"],4), "evals/pairs", round(d["config"]["column_density_evaluations_per_step_rank0"]/d["config"]["raytrace_updates_per_step"],3))
PY
done; done
