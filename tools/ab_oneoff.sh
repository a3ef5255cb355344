# Scratch driver for one-off A/B runs on the GPU box (edited per experiment; results are recorded in profiles/r02_diag_*.txt).
# This version: sub-box sweep (libc2ray.raytracing.do_all_sources semantics) with prebuilt variants of the library.
cp pyc2ray_amd/lib/libasora_hip.so build/variants/libasora_default_saved.so
trap 'cp build/variants/libasora_default_saved.so pyc2ray_amd/lib/libasora_hip.so' EXIT
for ROUND in 1 2; do for V in "$@"; do
  cp build/variants/libasora_$V.so pyc2ray_amd/lib/libasora_hip.so
  python tools/bench_c2ray_path.py --R 16 32 --cpu-sources 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$V round $ROUND R=%g call %.2f ms, sweep kernels %.3f ms' % (d['R'], d['s_per_call']*1e3, d['sweep_kernels_ms_per_call']))"
done; done
