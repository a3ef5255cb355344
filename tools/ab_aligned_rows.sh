# A/B of ASORA_OPT_ALIGNED_ROWS (rows of the rate grid cut at 64-byte lines): bench.py with the option forced off (15=1) and left to
# the library (on for the six-sector / twelve-pair units), alternating; then the atomic requests of both by counter.
# usage (GPU box): bash tools/ab_aligned_rows.sh "<R values>"
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
mkdir -p gpurun_out
for RR in ${1:-32}; do
  for round in 1 2; do
    for mode in off auto; do
      if [ $mode = off ]; then export PYC2RAY_AMD_OPTIONS="15=1"; else unset PYC2RAY_AMD_OPTIONS; fi
      timeout -k 10 300 python bench.py --steps 10 --warmup 3 --repeats 5 --cpu-sources 0 --R $RR 2>/dev/null > gpurun_out/aba.json || exit 1
      python -c "
import json;d=json.load(open('gpurun_out/aba.json'));k=d['kernels_ms_per_step']
print('R=$RR aligned rows $mode: step ms %.4f raytrace %.4f chemistry %.4f value %.4e evaluations/pairs %.4f' % (d['ms_per_step'], k['raytrace'], k['chemistry'], d['value'], d['config']['column_density_evaluations_per_step_rank0'] / d['config']['raytrace_updates_per_step']))"
    done
  done
done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for mode in off auto; do
  if [ $mode = off ]; then export PYC2RAY_AMD_OPTIONS="15=1"; else unset PYC2RAY_AMD_OPTIONS; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_aba_$mode -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sources 0 > $R/gpurun_out/pmc_aba_$mode.log 2>&1 || echo "pmc pass failed"
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_aba_$mode/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "raytrace_octant" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("R=32 aligned rows $mode:", {k: "%.5g (n=%d)" % (sum(v)/len(v), len(v)) for k,v in agg.items()})
PY
  rm -rf $R/gpurun_out/pmc_aba_$mode
done
