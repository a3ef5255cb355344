# Scratch driver for one-off A/B runs on the GPU box (edited per experiment; results are recorded in profiles/r02_diag_*.txt).
# This version: the diagnostics of r02_diag_raytrace_bottleneck.txt again, now that the work counters no longer serialise the launch
cp pyc2ray_amd/lib/libasora_hip.so build/variants/libasora_default_saved.so
trap 'cp build/variants/libasora_default_saved.so pyc2ray_amd/lib/libasora_hip.so' EXIT
for V in cur nodiv; do
  cp build/variants/libasora_$V.so pyc2ray_amd/lib/libasora_hip.so
  for RR in 16 32 64; do
    python bench.py --steps 6 --warmup 2 --cpu-sources 0 --R $RR 2>/dev/null > gpurun_out/one.json
    python -c "
import json;d=json.load(open('gpurun_out/one.json'));print('$V R=$RR raytrace_ms=%.4f'%(d['kernels_ms_per_step']['raytrace']))"
  done
done
cp build/variants/libasora_abl.so pyc2ray_amd/lib/libasora_hip.so
for RR in 16 32; do for A in 0 1 2; do
    ASORA_ABLATE=$A python bench.py --steps 6 --warmup 2 --cpu-sources 0 --R $RR 2>/dev/null > gpurun_out/one.json
    python -c "
import json;d=json.load(open('gpurun_out/one.json'));print('ablate=$A R=$RR raytrace_ms=%.4f'%(d['kernels_ms_per_step']['raytrace']))"
done; done
for RR in 16 32; do for X in 0 8000 18000 32000 72000; do
    ASORA_DIAG_EXTRA_LDS=$X python bench.py --steps 6 --warmup 2 --cpu-sources 0 --R $RR 2>/dev/null > gpurun_out/one.json
    python -c "
import json;d=json.load(open('gpurun_out/one.json'));print('extra_lds=$X R=$RR raytrace_ms=%.4f'%(d['kernels_ms_per_step']['raytrace']))"
done; done
