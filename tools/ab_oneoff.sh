# Scratch driver for one-off A/B runs on the GPU box (edited per experiment; results are recorded in profiles/r02_diag_*.txt).
# This version: the work counters on two addresses (cur) against spread over 4096 slots (slots)
cp pyc2ray_amd/lib/libasora_hip.so build/variants/libasora_default_saved.so
trap 'cp build/variants/libasora_default_saved.so pyc2ray_amd/lib/libasora_hip.so' EXIT
for ROUND in 1 2; do for V in cur slots; do
  cp build/variants/libasora_$V.so pyc2ray_amd/lib/libasora_hip.so
  for RR in 1 6 12 16 24 32 64; do
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-sources 0 --R $RR 2>/dev/null > gpurun_out/one.json
    python -c "
import json;d=json.load(open('gpurun_out/one.json'));print('$V round $ROUND R=$RR raytrace_ms=%.4f step_ms=%.4f'%(d['kernels_ms_per_step']['raytrace'], d['ms_per_step']))"
  done
done; done
