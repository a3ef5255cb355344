# GPU run 5: suite with the late-lookup kernel, evolve profile, headline numbers + stats + PMC for profiles/
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2e; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python tools/profile_evolve.py > $O/profile_evolve_sync.json 2> $O/profile_evolve.err
timeout -k 10 300 python tools/profile_evolve.py --sync 0 > $O/profile_evolve_nosync.json 2>> $O/profile_evolve.err
timeout -k 10 300 python tools/test1_stromgren.py --cpu-steps 0 > $O/test1.json 2> $O/test1.err
bash tools/measure_r2.sh e > $O/measure.log 2>&1
bash tools/pmc.sh r2 > $O/pmc.log 2>&1; cp gpurun_out/pmc_r2_summary.txt $O/
cat $O/profile_evolve_sync.json $O/profile_evolve_nosync.json
