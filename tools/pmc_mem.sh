# usage: bash tools/pmc_mem.sh TAG [bench.py arguments...]  -- like tools/pmc.sh, for the vector-memory pipeline (TA, TCP = L1, TCC = L2):
# request counts, summed latencies (mean latency = LATENCY / REQ, in cycles) and stall cycles, one counter group per pass
# (FETCH_SIZE and WRITE_SIZE in passes of their own, never combined with trace domains); per-kernel means of every counter
# land in gpurun_out/pmc_TAG_summary.txt.  ASORA_ABLATE in the environment is passed through (diagnostic builds only).
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}; TAG=$1; shift
export ASORA_ABLATE=${ASORA_ABLATE:-0}
cd /tmp
i=0
for C in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
         "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
         "TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" \
         "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_ATOMIC_sum" \
         "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcmem_${TAG}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sources 0 --evolving-state 0 "$@" > $R/gpurun_out/pmcmem_${TAG}_$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - <<PY
import csv,glob,collections
out=open("$R/gpurun_out/pmcmem_${TAG}_summary.txt","w")
out.write("# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 2 --warmup 1 --cpu-sources 0 --evolving-state 0 $*\n")
out.write("# kernel  counter  launches  mean per launch (FETCH_SIZE / WRITE_SIZE in KiB)\n")
for d in sorted(glob.glob("$R/gpurun_out/pmcmem_${TAG}_[0-9]*/")):
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
rm -rf $R/gpurun_out/pmcmem_${TAG}_[0-9]*
