# usage: bash tools/pmc_one.sh "<counters>" <bench args...>   -- one PMC pass, prints raytrace-kernel means
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; C="$1"; shift
cd /tmp; rm -rf $R/gpurun_out/pmc_one
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_one -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sources 0 --evolving-state 0 "$@" > $R/gpurun_out/pmc_one.log 2>&1
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$R/gpurun_out/pmc_one/*/*counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "raytrace" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: "%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
PY
