# Diagnostic: raytrace time against the number of workgroups a CU holds, by padding each workgroup's LDS (ablation build only).
# usage: bash tools/diag_occupancy.sh "<bench args>" "0 20000 45000 ..."
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
trap 'make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc > /dev/null 2>&1' EXIT
make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc EXTRA=-DASORA_ENABLE_ABLATION > /dev/null 2>&1
for X in $2; do
  ASORA_DIAG_EXTRA_LDS=$X timeout -k 10 300 python bench.py --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 $1 2>/dev/null > gpurun_out/occ.json
  python -c "
import json;d=json.load(open('gpurun_out/occ.json'));print('$1 extra LDS $X: raytrace ms', round(d['kernels_ms_per_step']['raytrace'],4))"
done
