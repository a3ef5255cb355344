# Attribute raytrace time with the PREBUILT diagnostic library (build/variants/libasora_abl.so, made in the build container with
# -DASORA_ENABLE_ABLATION; results are WRONG while a bit is set): bash tools/ablate_prebuilt.sh "0 1 2 4 5" "--R 16" "--R 16 --nsrc 8000"
#   1 = no rate atomics, 2 = no rates, 4 = no shell barriers
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
export PYC2RAY_AMD_LIBASORA=$PWD/build/variants/libasora_abl.so      # the production library is never overwritten
SET=$1; shift
for ARGS in "$@"; do for A in $SET; do
  ASORA_ABLATE=$A timeout -k 10 300 python bench.py --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --evolving-state 0 $ARGS 2>/dev/null > gpurun_out/abl.json
  python -c "
import json;d=json.load(open('gpurun_out/abl.json'));print('$ARGS ablate $A raytrace ms', round(d['kernels_ms_per_step']['raytrace'],4))"
done; done
