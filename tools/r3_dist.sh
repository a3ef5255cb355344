# Round-3 multi-rank records on ONE GPU (the build box has one): rehearsal of bench.py --gpus 2 / 4 over gloo (the box allows 6 processes on its GPU: launcher + ranks), the per-rank
# compute measurement + exchange model.  usage (GPU box): bash tools/r3_dist.sh
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; O=$R/gpurun_out/r3dist; mkdir -p $O; cd $R
# (tests/test_dist_gloo.py -m gpu runs with the rest of the GPU suite)
export PYC2RAY_AMD_BENCH_BACKEND=gloo PYC2RAY_AMD_BENCH_DEVICE=0
for P in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port $((29540+P)) bench.py --gpus $P --steps 3 --warmup 1 --repeats 2 > $O/rehearse$P.json 2> $O/rehearse$P.err; echo "rehearse$P exit $?"
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 3 --warmup 1 --repeats 2 --exchange allreduce > $O/rehearse2_allreduce.json 2> $O/rehearse2_allreduce.err; echo "rehearse2 allreduce exit $?"
unset PYC2RAY_AMD_BENCH_BACKEND PYC2RAY_AMD_BENCH_DEVICE
timeout -k 10 600 python tools/slab_compute_model.py --workload cosmo > $O/slab_compute_model_cosmo.json 2> $O/model_cosmo.err; echo "model cosmo exit $?"
timeout -k 10 600 python tools/slab_compute_model.py --workload uniform > $O/slab_compute_model_uniform.json 2> $O/model_uniform.err; echo "model uniform exit $?"
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/rehearse*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ranks", d["n_gpus"], "value", d["value"], "ms", d["ms_per_step"], "agree", d["config"]["ranks_agree_on_rates_and_ionised_fraction"], d["config"]["exchange_model"]["choice"] if d["config"]["exchange_model"] else None, "fallback", d["config"]["exchange_fallback"])
    except Exception as e: print(f, "unreadable:", e)
for f in sorted(glob.glob("$O/slab_compute_model_*.json")):
    d=json.load(open(f))
    for r in d["rows"]: print(f.split("/")[-1], "P", r["ranks"], "compute ms", round(r["slowest_rank_compute_ms"],3), "step", {k: round(v,3) for k,v in r["modelled_step_ms_overlapped"].items()}, "speedup", {k: round(v,2) for k,v in r["modelled_speedup_overlapped"].items()})
PY
