# the fused chemistry pass with (round 2) and without (round 3) the store of the folded rates, alternating on ONE box
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
for rep in 1 2 3; do for V in 0 1; do
  if [ $V = 1 ]; then export ASORA_DIAG_STORE_PHI=1; else unset ASORA_DIAG_STORE_PHI; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --cpu-sources 0 > gpurun_out/abc.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/abc.json")); print("store_phi=$V", "chemistry ms", round(d["kernels_ms_per_step"]["chemistry"],4), "raytrace ms", round(d["kernels_ms_per_step"]["raytrace"],4), "step ms", round(d["ms_per_step"],4))
PY
done; done
