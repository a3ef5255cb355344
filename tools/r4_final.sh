# Round-4 record set: everything profiles/r04_* (bench lines, kernel stats, counters) is copied from.  usage (GPU box): bash tools/r4_final.sh [part]
#   part 1: GPU suite, bench lines, kernel stats;  part 2: PMC passes;  part 3: the tools (c2ray path, geometry, PCIe, paper protocol, time steps, microbenchmarks, rehearsals)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; O=$R/gpurun_out/r4z; mkdir -p $O; cd $R
PART=${1:-123}
if [[ $PART == *1* ]]; then
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_uniform_R32.json 2> $O/bench_uniform_R32.err; echo "bench exit $?"
for RR in 16 64; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --R $RR --cpu-sources 0 > $O/bench_uniform_R$RR.json 2> $O/bench_uniform_R$RR.err; done
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --workload cosmo --cpu-sources 0 > $O/bench_cosmo_R32.json 2> $O/bench_cosmo_R32.err
timeout -k 10 600 python bench.py --N 512 --nsrc 100000 --workload cosmo --steps 3 --warmup 1 --repeats 3 --cpu-sources 0 > $O/bench_cfg4_512_1e5.json 2> $O/cfg4.err; echo "cfg4 exit $?"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 2 --cpu-sources 0 --evolving-state 0 > $O/stats.log 2>&1; echo "stats exit $?"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats16 -- python3 $R/bench.py --steps 10 --warmup 3 --repeats 1 --cpu-sources 0 --R 16 > $O/stats16.log 2>&1
cp $(find $O/stats16 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_R16.csv; rm -rf $O/stats16
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsc -- python3 $R/tools/time_subbox_device.py --R 32 --pairs 0 --reps 5 > $O/statsc.log 2>&1
cp $(find $O/statsc -name "*kernel_stats.csv" | head -1) $O/kernel_stats_subbox_device.csv; rm -rf $O/statsc
cd $R
fi
if [[ $PART == *2* ]]; then
bash tools/pmc.sh r4 > $O/pmc.log 2>&1; cp gpurun_out/pmc_r4_summary.txt $O/
bash tools/pmc.sh r4_R16 --R 16 > $O/pmc16.log 2>&1; cp gpurun_out/pmc_r4_R16_summary.txt $O/
bash tools/pmc_c2ray.sh > $O/pmc_c2ray.log 2>&1; cp gpurun_out/pmc_c2ray_summary.txt $O/
fi
if [[ $PART == *3* ]]; then
timeout -k 10 300 python tools/time_subbox_device.py > $O/subbox_device.jsonl 2> $O/subbox_device.err
timeout -k 10 600 python tools/bench_c2ray_path.py --R 16 32 --cpu-sources 1000 > $O/c2ray_path.jsonl 2> $O/c2ray.err; echo "c2ray path exit $?"
timeout -k 10 300 python tools/time_geometry_build.py --N 256 --nsrc 1000 > $O/time_geometry.jsonl 2> $O/geom.err; echo "geometry exit $?"
timeout -k 10 300 python tools/pcie_inclusive.py > $O/pcie_inclusive.json 2> $O/pcie.err
timeout -k 10 600 python tools/paper_benchmark.py > $O/paper_protocol.json 2> $O/paper.err
timeout -k 10 300 python tools/test1_stromgren.py --cpu-steps 0 > $O/test1.json 2> $O/test1.err
timeout -k 10 300 python tools/time_steps_resident.py > $O/time_steps_resident.json 2> $O/tsr.err
hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o /tmp/atomic_rate tools/micro/atomic_rate.hip 2> /dev/null && { for ROW in 8 24 40 64 4096; do /tmp/atomic_rate $ROW; done; for EDGE in 64 128 192 320 400; do /tmp/atomic_rate 40 $EDGE; done; } > $O/atomic_rate_microbench.txt 2>&1
export PYC2RAY_AMD_BENCH_BACKEND=gloo PYC2RAY_AMD_BENCH_DEVICE=0
for P in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port $((29540+P)) bench.py --gpus $P --steps 3 --warmup 1 --repeats 2 > $O/rehearse$P.json 2> $O/rehearse$P.err; echo "rehearse$P exit $?"
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 3 --warmup 1 --repeats 2 --exchange allreduce > $O/rehearse2_allreduce.json 2> $O/rehearse2_allreduce.err; echo "rehearse2 allreduce exit $?"
unset PYC2RAY_AMD_BENCH_BACKEND PYC2RAY_AMD_BENCH_DEVICE
fi
ls -la $O
