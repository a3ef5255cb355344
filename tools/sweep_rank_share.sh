#!/bin/bash
# Launch shapes for the source count ONE RANK of an 8-GPU run traces (configs[3]: 1000 sources in total -> 125 per rank): the raytrace
# kernel's mean launch for every decomposition x pairing, one fresh process each.   usage: tools/sweep_rank_share.sh [nsrc] [R]
NS=${1:-125}; R=${2:-32}
cd "$(dirname "$0")/.."
for S in 0 9 3 2 1; do for P in 0 1 2; do for T in 0 128; do
  [[ $S == 0 && ( $P != 0 || $T != 0 ) ]] && continue
  [[ $S != 0 && $P == 0 ]] && continue
  python bench.py --workload cosmo --nsrc $NS --R $R --sectors $S --pair-sources $P --block-threads $T --cpu-sources 0 --evolving-state 0 --repeats 3 --steps 10 2>/dev/null | S=$S P=$P T=$T python -c "
import json,os,sys
d=json.loads(sys.stdin.readline()); v=d['config']['raytrace_variant']
print('sectors %s pair %s threads %s -> units %d x %d threads, %s source(s) per workgroup%s: raytrace %.4f ms, step %.4f ms'
      % (os.environ['S'], os.environ['P'], os.environ['T'], v['units'], v['threads'], 2 if v['paired'] else 1, ', aligned' if v['aligned'] else '', d['roofline']['avg_launch_ms'], d['ms_per_step']))"
done; done; done
