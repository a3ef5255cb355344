# Round-2 record set: everything profiles/r02_* is copied from.  usage (GPU box): bash tools/r2_final.sh
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; O=$R/gpurun_out/r2z; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
bash tools/measure_r2.sh z > $O/measure.log 2>&1; cp gpurun_out/r2z/*.json gpurun_out/r2z/kernel_stats.csv $O/ 2>/dev/null
bash tools/pmc.sh r2 > $O/pmc.log 2>&1; cp gpurun_out/pmc_r2_summary.txt $O/
bash tools/pmc.sh r2_R16 --R 16 > $O/pmc16.log 2>&1; cp gpurun_out/pmc_r2_R16_summary.txt $O/
timeout -k 10 300 python tools/pcie_inclusive.py > $O/pcie_inclusive.json 2> $O/pcie.err
timeout -k 10 600 python tools/paper_benchmark.py > $O/paper_protocol.json 2> $O/paper.err
timeout -k 10 300 python tools/test1_stromgren.py --cpu-steps 1 > $O/test1.json 2> $O/test1.err
timeout -k 10 600 python tools/bench_c2ray_path.py --R 16 32 --cpu-sources 1000 > $O/c2ray_path.jsonl 2> $O/c2ray.err
timeout -k 10 600 python bench.py --N 512 --nsrc 100000 --workload cosmo --steps 3 --warmup 1 --cpu-sources 0 > $O/bench_cfg4_512_1e5.json 2> $O/cfg4.err
export PYC2RAY_AMD_BENCH_BACKEND=gloo PYC2RAY_AMD_BENCH_DEVICE=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err; echo "rehearse2 exit $?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 4 --steps 3 --warmup 1 > $O/rehearse4.json 2> $O/rehearse4.err; echo "rehearse4 exit $?"
ls -la $O
