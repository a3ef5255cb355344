# raytrace time for FEWER sources (what a rank of a multi-GPU run traces) x decomposition x workgroup size
# usage: bash tools/sweep_counts.sh "125 250 500" "0:0 9:256 9:512 3:256 3:512" [R]
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
mkdir -p gpurun_out
RR=${3:-32}
for NS in $1; do for ST in $2; do
  S=${ST%%:*}; T=${ST##*:}
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --R $RR --nsrc $NS --sectors $S --block-threads $T --pair-sources ${PAIRS:-0} > gpurun_out/swc.json 2>/dev/null || { echo "nsrc=$NS sectors=$S threads=$T FAILED"; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/swc.json")); print("R=$RR nsrc=$NS sectors=$S threads=$T pairs=${PAIRS:-0}", "raytrace ms", round(d["kernels_ms_per_step"]["raytrace"],4))
PY
done; done
