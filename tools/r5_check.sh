# Round-5 intermediate check on the GPU box: bash tools/r5_check.sh OUTDIR [parts]   (parts: t = tests, w = world-1 RCCL slab path, n = N = 576 families, d = de-phase experiment)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
PART=${2:-twnd}
if [[ $PART == *t* ]]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
fi
if [[ $PART == *w* ]]; then
for i in 1 2; do
  PYC2RAY_AMD_FORCE_COLLECTIVE=1 timeout -k 10 300 python bench.py --gpus 1 --workload cosmo --steps 20 --warmup 5 --cpu-sources 0 > $O/world1_slab_$i.json 2> $O/world1_slab_$i.err; echo "world-1 slab rc=$?"
done
PYC2RAY_AMD_FORCE_COLLECTIVE=1 timeout -k 10 300 python bench.py --gpus 1 --workload cosmo --steps 20 --warmup 5 --cpu-sources 0 --exchange allreduce > $O/world1_allreduce.json 2> $O/world1_allreduce.err; echo "world-1 allreduce rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --workload cosmo --steps 20 --warmup 5 --cpu-sources 0 > $O/cosmo_1gpu.json 2> $O/cosmo_1gpu.err
PYC2RAY_AMD_BENCH_BACKEND=gloo PYC2RAY_AMD_BENCH_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --exchange slab > $O/selflaunch_2ranks_gloo.json 2> $O/selflaunch_2ranks_gloo.err; echo "self-launched 2 ranks rc=$?"
fi
if [[ $PART == *n* ]]; then
timeout -k 10 300 python bench.py --N 576 --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --evolving-state 0 > $O/bench_576.json 2> $O/bench_576.err; echo "576 rc=$?"
PYC2RAY_AMD_OPTIONS="12=1" timeout -k 10 300 python bench.py --N 576 --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --evolving-state 0 > $O/bench_576_global_atomics.json 2> $O/bench_576_global_atomics.err; echo "576 global rc=$?"
fi
if [[ $PART == *d* ]]; then
timeout -k 10 300 python tools/dephase_small_R.py > $O/dephase_R16.json 2> $O/dephase_R16.err; echo "dephase rc=$?"
timeout -k 10 300 python tools/dephase_small_R.py --R 12 > $O/dephase_R12.json 2> $O/dephase_R12.err
fi
ls $O
