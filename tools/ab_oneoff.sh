# Scratch driver for one-off A/B runs on the GPU box (edited per experiment; the experiments it ran are recorded in
# profiles/r02_diag_*.txt).  Pattern: swap in a prebuilt variant, loop over settings, print one line per run.
cp pyc2ray_amd/lib/libasora_hip.so build/variants/libasora_default_saved.so
trap 'cp build/variants/libasora_default_saved.so pyc2ray_amd/lib/libasora_hip.so' EXIT
cp build/variants/libasora_${1:-abl}.so pyc2ray_amd/lib/libasora_hip.so
shift
for RR in "$@"; do
    python bench.py --steps 6 --warmup 2 --cpu-sources 0 --R $RR 2>/dev/null > gpurun_out/one.json
    python -c "
import json;d=json.load(open('gpurun_out/one.json'));print('R=$RR raytrace_ms=%.4f step_ms=%.4f'%(d['kernels_ms_per_step']['raytrace'], d['ms_per_step']))"
done
