# A/B of prebuilt library variants (build/variants/libasora_<name>.so) on the default bench job, alternating on ONE box.
#   bash tools/ab_variants.sh "ntloads nts1 nts2" [rounds]
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
mkdir -p gpurun_out
for ROUND in $(seq 1 ${2:-2}); do
for V in $1; do
    PYC2RAY_AMD_LIBASORA=$PWD/build/variants/libasora_$V.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --repeats 3 --cpu-sources 0 --evolving-state 0 > gpurun_out/abv.json 2>/dev/null || { echo "$V FAILED"; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/abv.json")); k=d["kernels_ms_per_step"]
print("round $ROUND %-12s ms/step %.4f  raytrace %.4f  fused pass %.4f" % ("$V", d["ms_per_step"], k["raytrace"], k["chemistry"]))
PY
done
done
