# paired-sources variant with and without a register bound (ASORA_PAIR_MIN_WAVES): bash tools/ab_pairs_build.sh 16 24 32 48
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
trap 'make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc > /dev/null 2>&1' EXIT
one() { # R pair threads
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --R $1 --pair-sources $2 --block-threads $3 > gpurun_out/abp.json 2>/dev/null || { echo "R=$1 pair=$2 threads=$3 FAILED"; return; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abp.json")); print("R=$1 pair=$2 threads=$3", "raytrace ms", round(d["kernels_ms_per_step"]["raytrace"],4), "step ms", round(d["ms_per_step"],4))
PY
}
for W in 1 4; do
  make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc EXTRA=-DASORA_PAIR_MIN_WAVES=$W > /dev/null 2>&1
  echo "== ASORA_PAIR_MIN_WAVES=$W"
  for RR in "$@"; do
    one $RR 1 0
    for T in 0 64 128 256; do one $RR 2 $T; done
  done
done
