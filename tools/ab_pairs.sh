# A/B of ASORA_OPT_PAIR_SOURCES (two sources per workgroup) on ONE box: bash tools/ab_pairs.sh 16 24 32 48 64
# per radius: the default kernel, then the paired variant in the default launch shape and with other workgroup sizes
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
mkdir -p gpurun_out
one() { # R pair threads sectors
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --R $1 --pair-sources $2 --block-threads $3 --sectors $4 > gpurun_out/abp.json 2>/dev/null || { echo "R=$1 pair=$2 threads=$3 sectors=$4 FAILED"; return; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abp.json")); print("R=$1 pair=$2 threads=$3 sectors=$4", "raytrace ms", round(d["kernels_ms_per_step"]["raytrace"],4), "step ms", round(d["ms_per_step"],4))
PY
}
for RR in "$@"; do
  one $RR 1 0 0
  one $RR 2 0 0
  for T in 64 128 256; do one $RR 2 $T 0; done
  one $RR 1 0 0
done
