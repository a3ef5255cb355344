# Diagnostic: attribute raytrace-kernel time by switching parts off (results are WRONG while a bit is set).
# Rebuilds the library with -DASORA_ENABLE_ABLATION, runs bench.py per ASORA_ABLATE value, restores the build.
#   1 = no rate atomics, 2 = no rates, 4 = no shell barriers
# whatever happens, leave the default build behind (a diagnostic build gives WRONG results under ASORA_ABLATE)
trap 'make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc > /dev/null 2>&1' EXIT
make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc EXTRA=-DASORA_ENABLE_ABLATION > /dev/null 2>&1
for A in ${ABLATE_SET:-0 1 2}; do
  ASORA_ABLATE=$A timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sources 0 $BENCH_ARGS 2>/dev/null > gpurun_out/abl_$A.json
  python -c "
import json;d=json.load(open('gpurun_out/abl_$A.json'));print('ablate $A', d['ms_per_step'], d['kernels_ms_per_step']['raytrace'])"
done
make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc > /dev/null 2>&1
