# GPU run 6: uniform-temperature chemistry pass, kernel-level profile of the small-problem loop
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2f; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-sources 0 > $O/bench_R32_$i.json 2> $O/bench.err; python -c "
import json;d=json.load(open('$O/bench_R32_$i.json'));print('R32', d['ms_per_step'], d['kernels_ms_per_step'])"; done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/tools/profile_evolve.py --sync 0 > $O/stats1.log 2>&1
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/test1_kernel_stats.csv; rm -rf $O/stats1
cat $O/test1_kernel_stats.csv
