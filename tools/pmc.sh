# usage: bash tools/pmc.sh TAG [bench.py arguments...]  -- PMC passes of bench.py (2 timed steps), one counter group per pass
# (FETCH_SIZE and WRITE_SIZE in passes of their own, never combined with trace domains); per-kernel means of every counter
# land in gpurun_out/pmc_TAG_summary.txt.  ASORA_ABLATE in the environment is passed through (diagnostic builds only).
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; TAG=$1; shift
export ASORA_ABLATE=${ASORA_ABLATE:-0}
cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
         "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sources 0 --evolving-state 0 "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - <<PY
import csv,glob,collections,ctypes,os
out=open("$R/gpurun_out/pmc_${TAG}_summary.txt","w")
# the build of the library these counters belong to (asora_build_id: hash of its sources, headers and flags); bench.py
# only pairs a summary with timings of the same build
_l=ctypes.CDLL(os.environ.get("PYC2RAY_AMD_LIBASORA") or "$R/pyc2ray_amd/lib/libasora_hip.so"); _l.asora_build_id.restype=ctypes.c_char_p
out.write("# build_id %s\n" % _l.asora_build_id().decode())
out.write("# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 2 --warmup 1 --cpu-sources 0 --evolving-state 0 $*\n")
out.write("# kernel  counter  launches  mean per launch (FETCH_SIZE / WRITE_SIZE in KiB)\n")
for d in sorted(glob.glob("$R/gpurun_out/pmc_${TAG}_[0-9]*/")):
    for f in glob.glob(d+"*/*counter_collection.csv"):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(list)
        for r in rows:
            name=r["Kernel_Name"].split("(")[0].replace("void ","").replace("asora::","").split("<")[0]
            agg[(name,r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k,c),v in sorted(agg.items()):
            if "raytrace" in k or "chemistry" in k or "prepare" in k or "subbox" in k:
                line=f"{k:42s} {c:26s} n={len(v)} mean={sum(v)/len(v):.5g}"
                print(line); out.write(line+"\n")
PY
rm -rf $R/gpurun_out/pmc_${TAG}_[0-9]*
