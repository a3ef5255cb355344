# Launch shapes at r_RT = 16, 1000 sources (BASELINE configs[2] sweep point): decomposition x pairing x workgroup size, one fresh process each.  usage (GPU box): bash tools/sweep_shapes_R16.sh
cd $GRAFT_REPO_ROOT
for S in 0 6 7 8 9 3; do for P in 1 2; do for T in 0 256 512; do
  [[ $S == 0 && ( $P != 1 || $T != 0 ) ]] && continue
  PP=$P; [[ $S == 0 ]] && PP=0
  python bench.py --R 16 --sectors $S --pair-sources $PP --block-threads $T --cpu-sources 0 --evolving-state 0 --repeats 3 --steps 10 2>/dev/null | S=$S P=$PP T=$T python -c "
import json,os,sys
d=json.loads(sys.stdin.readline()); v=d['config']['raytrace_variant']
print('sectors %s pair %s threads %s -> units %d x %d threads, %s source(s) per workgroup%s: raytrace %.4f ms frac %.3f, step %.4f ms'
      % (os.environ['S'], os.environ['P'], os.environ['T'], v['units'], v['threads'], 2 if v['paired'] else 1, ', aligned' if v['aligned'] else '', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step']))"
done; done; done
