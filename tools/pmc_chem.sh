# usage: bash tools/pmc_chem.sh  -- VALU instruction and wave-cycle counters of every launch of the fused chemistry pass in
# tools/chem_front.py (quiet medium first, then the medium with fronts), in launch order -> gpurun_out/pmc_chem_front.txt
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}
cd /tmp; rm -rf $R/gpurun_out/pmc_chem
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_chem -- python3 $R/tools/chem_front.py --histogram 0 > $R/gpurun_out/pmc_chem.log 2>&1 || echo "pmc pass failed"
python3 - <<PY
import csv,glob,collections
out=open("$R/gpurun_out/pmc_chem_front.txt","w")
out.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES -- python3 tools/chem_front.py --histogram 0\n")
out.write("# launches of chemistry_tile_kernel in order (the run does a quiet medium first, then the medium with fronts; see the JSON of the tool)\n")
for f in glob.glob("$R/gpurun_out/pmc_chem/*/*counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "chemistry_tile_kernel" in r["Kernel_Name"]]
    by=collections.OrderedDict()
    for r in rows: by.setdefault(int(r["Dispatch_Id"]),{})[r["Counter_Name"]]=float(r["Counter_Value"])
    for n,(d,c) in enumerate(by.items()):
        line=f"launch {n:3d} "+" ".join(f"{k}={v:.5g}" for k,v in sorted(c.items()))
        print(line); out.write(line+"\n")
PY
rm -rf $R/gpurun_out/pmc_chem
