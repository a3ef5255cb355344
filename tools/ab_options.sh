#!/bin/bash
# Alternating A/B of library OPTIONS (and/or library builds) on ONE box: every arm is a fresh process of bench.py, the arms take
# turns for ROUNDS rounds, one line per run with the raytrace kernel's mean launch on the quiet benchmark medium and on the
# evolving field (bench.py `evolving_state`), the step and the fused pass.
#   usage: tools/ab_options.sh ROUNDS "bench.py arguments" label[=OPTIONS[@LIBRARY]] ...
#   OPTIONS: the value of PYC2RAY_AMD_OPTIONS for that arm ("5=3" = twelve sector pairs forced, "" = the library's choices);
#   LIBRARY: a diagnostic build to load instead of pyc2ray_amd/lib/libasora_hip.so (PYC2RAY_AMD_LIBASORA)
#   e.g.   tools/ab_options.sh 2 "--numtau 2000 --R 32" base u12=5=3
ROUNDS=$1; ARGS=$2; shift 2
cd "$(dirname "$0")/.."
for r in $(seq $ROUNDS); do
  for arm in "$@"; do
    label=${arm%%=*}; rest=""; [[ "$arm" == *=* ]] && rest=${arm#*=}
    opts=${rest%%@*}; lib=""; [[ "$rest" == *@* ]] && lib=${rest#*@}
    PYC2RAY_AMD_OPTIONS="$opts" PYC2RAY_AMD_LIBASORA="$lib" python bench.py $ARGS --evolving-state 1 --cpu-sources 0 --repeats 3 2>/dev/null | LABEL="$label" ROUND=$r python -c "
import json,os,sys
d=json.loads(sys.stdin.readline()); e=d['evolving_state']
print('%s round %s | quiet raytrace_ms %.4f frac %.3f | evolving raytrace_ms %.4f frac %.3f (%d iterations) | step %.4f | pass %.4f | variant units %s threads %s'
      % (os.environ['LABEL'], os.environ['ROUND'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], e['raytrace_ms_mean'], e['roofline_frac'], e['outer_iterations'],
         d['ms_per_step'], d['kernels_ms_per_step']['chemistry'], d['config'].get('raytrace_variant', {}).get('units'), d['config'].get('raytrace_variant', {}).get('threads')))"
  done
done
