# Scratch driver for one-off A/B runs on the GPU box (edited per experiment; results are recorded in profiles/r02_*.txt).
# This version: prebuilt variants, 20 timed steps each, four rounds, radii as arguments after the variant list
NAMES=$1; shift
for ROUND in 1 2 3 4; do for V in $NAMES; do
  export PYC2RAY_AMD_LIBASORA=$PWD/build/variants/libasora_$V.so
  for RR in "$@"; do
    python bench.py --steps 20 --warmup 5 --cpu-sources 0 --R $RR 2>/dev/null > gpurun_out/one.json
    python -c "
import json;d=json.load(open('gpurun_out/one.json'));print('$V round $ROUND R=$RR raytrace_ms=%.4f'%(d['kernels_ms_per_step']['raytrace']))"
  done
done; done
