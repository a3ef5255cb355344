#!/bin/bash
# A/B of the rate tables in LDS (ASORA_OPT_LDS_TABLES = option 18) on ONE box, alternating processes:
#   base   the library's choice (six sectors, tables in global memory)
#   u12    twelve sector pairs forced (option 5 = 3), tables in global memory -- what the shape change alone does
#   lds    tables in LDS (which takes the twelve sector pairs by itself)
# workload: bench.py --numtau 2000 (the reference's production table size) with --evolving-state 1; prints per run the raytrace
# kernel's mean launch on the quiet benchmark medium and on the evolving field.    usage: tools/ab_lds_tables.sh [rounds] [R]
ROUNDS=${1:-2}; R=${2:-32}
cd "$(dirname "$0")/.."
for r in $(seq $ROUNDS); do
  for v in base u12 lds; do
    case $v in base) export PYC2RAY_AMD_OPTIONS="";; u12) export PYC2RAY_AMD_OPTIONS="5=3";; lds) export PYC2RAY_AMD_OPTIONS="18=2";; esac
    python bench.py --numtau 2000 --R $R --evolving-state 1 --cpu-sources 0 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d['evolving_state']
print('$v round $r R=$R quiet raytrace_ms %.4f frac %.3f | evolving raytrace_ms %.4f frac %.3f (%d iterations) | step %.4f | pass %.4f'
      % (d['roofline']['avg_launch_ms'], d['roofline']['frac'], e['raytrace_ms_mean'], e['roofline_frac'], e['outer_iterations'], d['ms_per_step'], d['kernels_ms_per_step']['chemistry']))"
  done
done
