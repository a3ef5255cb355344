# A/B of this round's library against round 4's on ONE box: round 4's tree (commit 9b9afea) as a git worktree under build/r04src with its
# own library built there (build container: `git worktree add -f build/r04src 9b9afea && make -C build/r04src/pyc2ray_amd/csrc -j4`).
# usage (GPU box): bash tools/ab_vs_r04.sh OUTDIR
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; O=$R/gpurun_out/$1; mkdir -p $O
for ROUND in 1 2; do
for WHO in r04 r05; do
  if [ $WHO = r04 ]; then cd $R/build/r04src; else cd $R; fi
  for RR in 16 32; do
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --repeats 3 --cpu-sources 0 --evolving-state 0 --R $RR > $O/ab_${WHO}_R${RR}_$ROUND.json 2> $O/ab_${WHO}_R${RR}_$ROUND.err
    python -c "
import json;d=json.load(open('$O/ab_${WHO}_R${RR}_$ROUND.json'));k=d['kernels_ms_per_step'];print('$WHO round $ROUND R=$RR raytrace %.4f pass %.4f step %.4f'%(k['raytrace'],k['chemistry'],d['ms_per_step']))"
  done
  timeout -k 10 300 python tools/time_subbox_device.py --reps 10 $( [ $WHO = r05 ] && echo "--aligned 1" ) 2> /dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$WHO round $ROUND subbox R=%g pairs=%d round=%d %.4f ms'%(d['R'],d['pair_sources_option'],d['round'],d['sweep_kernel_ms_per_call']))"
done; done
