# Round-6 record set: everything profiles/r06_* of the "record set" kind (bench lines, kernel stats, counters, tools) is copied from here.
# usage (GPU box): bash tools/r6_final.sh [parts]
#   1: GPU suite, bench lines, kernel stats;  2: PMC passes (summaries carry the library's build id);  3: tools (sub-box path, geometry, PCIe,
#   paper protocol, application-level tests, time steps through the class);  4: multi-rank code paths on one GPU (world-1 RCCL, self-launched
#   gloo rehearsals with the configs4 block, per-rank compute models)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; O=$R/gpurun_out/r6z; mkdir -p $O; cd $R
PART=${1:-1234}
python -c "
import sys; sys.path.insert(0,'$R')
from pyc2ray_amd.load_extensions import load_asora; print('build_id', load_asora().build_id())" > $O/build_id.txt 2>/dev/null
if [[ $PART == *1* ]]; then
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_uniform_R32.json 2> $O/bench_uniform_R32.err; echo "bench exit $?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --numtau 2000 --evolving-state 1 --cpu-sources 0 > $O/bench_uniform_R32_numtau2000.json 2> $O/bench_numtau2000.err
for RR in 16 64; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --R $RR --cpu-sources 0 > $O/bench_uniform_R$RR.json 2> $O/bench_uniform_R$RR.err; done
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --workload cosmo --cpu-sources 0 > $O/bench_cosmo_R32.json 2> $O/bench_cosmo_R32.err
timeout -k 10 600 python bench.py --N 512 --nsrc 100000 --workload cosmo --steps 3 --warmup 1 --repeats 3 --cpu-sources 0 > $O/bench_cfg4_512_1e5.json 2> $O/cfg4.err; echo "cfg4 exit $?"
timeout -k 10 300 python bench.py --N 576 --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --evolving-state 0 > $O/bench_576.json 2> $O/bench_576.err
timeout -k 10 300 python bench.py --N 576 --R 16 --steps 10 --warmup 3 --repeats 3 --cpu-sources 0 --evolving-state 0 > $O/bench_576_R16.json 2> $O/bench_576_R16.err
tools/ab_options.sh 2 "--steps 20 --warmup 5" "base|||" "u12||--sectors 3|" > $O/ab_twelve_sector_pairs.txt 2>&1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 2 --cpu-sources 0 --evolving-state 0 > $O/stats.log 2>&1; echo "stats exit $?"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/stats
cd $R
fi
if [[ $PART == *2* ]]; then
bash tools/pmc.sh r6 > $O/pmc.log 2>&1; cp gpurun_out/pmc_r6_summary.txt $O/
fi
if [[ $PART == *3* ]]; then
timeout -k 10 300 python tools/time_subbox_device.py > $O/subbox_device.jsonl 2> $O/subbox_device.err
timeout -k 10 600 python tools/bench_c2ray_path.py --R 16 32 --cpu-sources 1000 > $O/c2ray_path.jsonl 2> $O/c2ray.err; echo "c2ray path exit $?"
timeout -k 10 300 python tools/time_geometry_build.py --N 128 256 320 > $O/time_geometry.jsonl 2> $O/geom.err; echo "geometry exit $?"
timeout -k 10 300 python tools/pcie_inclusive.py > $O/pcie_inclusive.json 2> $O/pcie.err
timeout -k 10 600 python tools/paper_benchmark.py > $O/paper_protocol.json 2> $O/paper.err
timeout -k 10 300 python tools/test1_stromgren.py --cpu-steps 0 > $O/test1.json 2> $O/test1.err
timeout -k 10 300 python tools/test2_cosmo_ifront.py > $O/test2.json 2> $O/test2.err; echo "test2 exit $?"
timeout -k 10 300 python tools/hackathon_test1.py > $O/hackathon_test1_128.json 2> $O/hackathon.err; echo "hackathon exit $?"
timeout -k 10 300 python tools/time_steps_resident.py --out $O/time_steps_resident.json > /dev/null 2> $O/tsr.err
timeout -k 10 300 python tools/time_steps_resident.py --cosmological 1 --out $O/time_steps_resident_cosmological.json > /dev/null 2>> $O/tsr.err
fi
if [[ $PART == *4* ]]; then
PYC2RAY_AMD_FORCE_COLLECTIVE=1 timeout -k 10 300 python bench.py --gpus 1 --workload cosmo --steps 20 --warmup 5 --cpu-sources 0 > $O/world1_rccl_slab_path.json 2> $O/world1_slab.err; echo "world-1 slab exit $?"
PYC2RAY_AMD_FORCE_COLLECTIVE=1 timeout -k 10 300 python bench.py --gpus 1 --workload cosmo --steps 20 --warmup 5 --cpu-sources 0 --exchange allreduce > $O/world1_rccl_allreduce_path.json 2> $O/world1_allreduce.err
export PYC2RAY_AMD_BENCH_BACKEND=gloo PYC2RAY_AMD_BENCH_DEVICE=0
for P in 2 4; do     # no launcher around them: bench.py starts its own ranks; the configs4 block runs at its full size
  timeout -k 10 600 python bench.py --gpus $P --steps 3 --warmup 1 --repeats 2 --exchange slab > $O/selflaunch_${P}ranks_gloo.json 2> $O/selflaunch_$P.err; echo "self-launched $P ranks exit $?"
done
timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --repeats 2 --exchange allreduce > $O/selflaunch_2ranks_allreduce_gloo.json 2> $O/selflaunch_2a.err
unset PYC2RAY_AMD_BENCH_BACKEND PYC2RAY_AMD_BENCH_DEVICE
timeout -k 10 300 python tools/slab_compute_model.py --chunks 1 > $O/slab_compute_model_cosmo.json 2> $O/slab_model.err; echo "slab model exit $?"
timeout -k 10 500 python tools/slab_compute_model.py --N 512 --nsrc 100000 --reps 2 --chunks 1 > $O/slab_compute_model_cfg4.json 2>> $O/slab_model.err
fi
ls -la $O
