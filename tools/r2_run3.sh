# GPU run 3 of round 2: fixed tests, fused-loop numbers, PMC for R=16/32, small-radius shape sweep, 2-rank rehearsal
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2c; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-sources 0 > $O/bench_R32.json 2> $O/bench_R32.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-sources 0 --R 16 > $O/bench_R16.json 2> $O/bench_R16.err
timeout -k 10 300 python tools/test1_stromgren.py --cpu-steps 0 > $O/test1.json 2> $O/test1.err
for S in 1 2 3; do for T in 64 128; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-sources 0 --R 16 --sectors $S --block-threads $T > $O/sw16_${S}_$T.json 2>/dev/null
  python -c "
import json;d=json.load(open('$O/sw16_${S}_$T.json'));print('R16 sectors $S threads $T', d['kernels_ms_per_step']['raytrace'])"
done; done
bash tools/pmc.sh r2_R16 --R 16 > $O/pmc_R16.log 2>&1
bash tools/pmc.sh r2_R32 > $O/pmc_R32.log 2>&1
cp gpurun_out/pmc_r2_R16_summary.txt gpurun_out/pmc_r2_R32_summary.txt $O/
export PYC2RAY_AMD_BENCH_BACKEND=gloo PYC2RAY_AMD_BENCH_DEVICE=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err; echo "rehearse2 exit $?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 4 --steps 3 --warmup 1 > $O/rehearse4.json 2> $O/rehearse4.err; echo "rehearse4 exit $?"
ls $O
