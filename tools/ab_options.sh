#!/bin/bash
# Alternating A/B of library options / bench arguments (and/or library builds) on ONE box: every arm is a fresh process of bench.py,
# the arms take turns for ROUNDS rounds, one line per run with the raytrace kernel's mean launch on the quiet benchmark medium and
# on the evolving field (bench.py `evolving_state`), the step, the fused pass and the launch shape the timed launches took.
#   usage: tools/ab_options.sh ROUNDS "common bench.py arguments" "label|PYC2RAY_AMD_OPTIONS|extra bench.py arguments|library" ...
#   (bench.py itself sets the options it has arguments for -- --sectors, --pair-sources, --block-threads, --z-transposed -- after
#    device_init: force THOSE through the arguments, everything else through PYC2RAY_AMD_OPTIONS; library: a diagnostic build to load
#    instead of pyc2ray_amd/lib/libasora_hip.so)
#   e.g.   tools/ab_options.sh 2 "--steps 20 --warmup 5" "base|||" "u12||--sectors 3|"
ROUNDS=$1; ARGS=$2; shift 2
cd "$(dirname "$0")/.."
for r in $(seq $ROUNDS); do
  for arm in "$@"; do
    IFS='|' read -r label opts extra lib <<< "$arm"
    PYC2RAY_AMD_OPTIONS="$opts" PYC2RAY_AMD_LIBASORA="$lib" python bench.py $ARGS $extra --evolving-state 1 --cpu-sources 0 --repeats 3 2>/dev/null | LABEL="$label" ROUND=$r python -c "
import json,os,sys
d=json.loads(sys.stdin.readline()); e=d['evolving_state']; v=d['config'].get('raytrace_variant', {})
print('%s round %s | quiet raytrace_ms %.4f frac %.3f | evolving raytrace_ms %.4f frac %.3f (%d iterations) | step %.4f | pass %.4f | %s units x %s threads, %s source(s) per workgroup%s'
      % (os.environ['LABEL'], os.environ['ROUND'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], e['raytrace_ms_mean'], e['roofline_frac'], e['outer_iterations'],
         d['ms_per_step'], d['kernels_ms_per_step']['chemistry'], v.get('units'), v.get('threads'), 2 if v.get('paired') else 1, ', line-aligned' if v.get('aligned') else ''))"
  done
done
