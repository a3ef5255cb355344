# Round-2 measurement set (run on the GPU box through gpurun).  Outputs under gpurun_out/r2/.
# usage: bash tools/measure_r2.sh [tag]
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}
TAG=${1:-a}; O=$R/gpurun_out/r2$TAG; mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_uniform_R32.json 2> $O/bench_uniform_R32.err && \
for RR in 16 64; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --R $RR --cpu-sources 0 > $O/bench_uniform_R$RR.json 2> $O/bench_uniform_R$RR.err || exit 1; done
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --workload cosmo --cpu-sources 0 > $O/bench_cosmo_R32.json 2> $O/bench_cosmo_R32.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-sources 0 > $O/stats.log 2>&1 || exit 1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
ls -la $O
