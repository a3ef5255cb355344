# one-off: diagnostic build, where do the memory-side requests hurt (64: rate atomics into a 512 KiB window, 128: nHI from a 512 KiB window)
cp pyc2ray_amd/lib/libasora_hip.so build/variants/libasora_default_saved.so
trap 'cp build/variants/libasora_default_saved.so pyc2ray_amd/lib/libasora_hip.so' EXIT
cp build/variants/libasora_abl.so pyc2ray_amd/lib/libasora_hip.so
for RR in 16 32; do for A in 0 64 128 192 1 129; do
    ASORA_ABLATE=$A python bench.py --steps 6 --warmup 2 --cpu-sources 0 --R $RR 2>/dev/null > gpurun_out/one.json
    python -c "
import json;d=json.load(open('gpurun_out/one.json'));print('R=$RR ablate=$A raytrace_ms=%.4f'%(d['kernels_ms_per_step']['raytrace']))"
done; done
