# GPU run 8: sub-box sweep after the rework (LDS / global shells), full suite
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2h; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python tools/bench_c2ray_path.py --R 16 32 --cpu-sources 0 > $O/c2ray_lds.jsonl 2> $O/c2ray.err
timeout -k 10 300 python tools/bench_c2ray_path.py --R 16 32 --cpu-sources 0 --global-shells 1 > $O/c2ray_global.jsonl 2>> $O/c2ray.err
cat $O/c2ray_lds.jsonl $O/c2ray_global.jsonl
timeout -k 10 300 python tools/test1_stromgren.py --cpu-steps 0 > $O/test1.json 2> $O/test1.err; cat $O/test1.json
