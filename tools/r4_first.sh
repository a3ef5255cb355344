# Round 4, first GPU call: the whole GPU suite on the tree with the ADVICE fixes, the bench lines (default, R = 16, 64), the
# rehearsal of bench.py --gpus 2 / 4 over gloo on the one GPU.  usage (GPU box): bash tools/r4_first.sh
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_uniform_R32.json 2> $O/bench_uniform_R32.err; echo "bench exit $?"
for RR in 16 64; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --R $RR --cpu-sources 0 > $O/bench_uniform_R$RR.json 2> $O/bench_uniform_R$RR.err; echo "bench R$RR exit $?"; done
export PYC2RAY_AMD_BENCH_BACKEND=gloo PYC2RAY_AMD_BENCH_DEVICE=0
for P in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port $((29540+P)) bench.py --gpus $P --steps 3 --warmup 1 --repeats 2 > $O/rehearse$P.json 2> $O/rehearse$P.err; echo "rehearse$P exit $?"
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 3 --warmup 1 --repeats 2 --exchange allreduce > $O/rehearse2_allreduce.json 2> $O/rehearse2_allreduce.err; echo "rehearse2 allreduce exit $?"
unset PYC2RAY_AMD_BENCH_BACKEND PYC2RAY_AMD_BENCH_DEVICE
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json"))+sorted(glob.glob("$O/rehearse*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "ranks", d["n_gpus"], "value %.4g" % d["value"], "ms %.4f" % d["ms_per_step"], "rt", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"],
              "1gpu", d.get("one_gpu_same_workload_ms_per_step"), "speedup", d.get("speedup_vs_one_gpu"), "phases", d.get("phases_ms"), "links", d.get("measured_link_GBs"))
        if "evolving_state" in d: print("  evolving:", {k: d["evolving_state"].get(k) for k in ("raytrace_ms_mean","fused_pass_ms_mean","fused_pass_ms_max","outer_iterations","failed")})
    except Exception as e: print(f, "unreadable:", e)
PY
