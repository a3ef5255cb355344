# A/B of the fused pass with non-temporal loads of its read streams (production) against plain loads, alternating on ONE box.
#   here:  make -C pyc2ray_amd/csrc EXTRA=-DASORA_CHEM_NT_LOADS=0 OUT=$PWD/build/variants/libasora_plainloads.so
#          cp pyc2ray_amd/lib/libasora_hip.so build/variants/libasora_ntloads.so
#   box:   bash tools/ab_chem_ntloads.sh
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
mkdir -p gpurun_out
for ROUND in 1 2 3; do
for V in plainloads ntloads; do
  for W in uniform cosmo; do
    PYC2RAY_AMD_LIBASORA=$PWD/build/variants/libasora_$V.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --repeats 3 --cpu-sources 0 --workload $W --evolving-state $([ $ROUND = 1 ] && [ $W = uniform ] && echo 1 || echo 0) > gpurun_out/abnt.json 2>/dev/null || { echo "$V $W FAILED"; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/abnt.json")); k=d["kernels_ms_per_step"]; e=d.get("evolving_state") or {}
print("round $ROUND $V $W: ms/step %.4f  raytrace %.4f  fused pass %.4f" % (d["ms_per_step"], k["raytrace"], k["chemistry"]), ("  evolving: raytrace %.4f pass %.4f" % (e["raytrace_ms_mean"], e["fused_pass_ms_mean"])) if e else "")
PY
  done
done
done
