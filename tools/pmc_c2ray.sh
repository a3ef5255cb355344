# usage: bash tools/pmc_c2ray.sh  -- HBM and atomic counters of the sub-box sweep kernels (raytrace_octant_kernel = the tabulated SUBBOX variant, round 3;
# subbox_sweep_kernel = the on-the-fly one, now only the dumped source) (libc2ray.raytracing.do_all_sources
# semantics, 1000 sources, 256^3, r_RT = 32), one rocprofv3 --pmc pass per group -> gpurun_out/pmc_c2ray_summary.txt
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}
cd /tmp; i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_c2ray_$i -- python3 $R/tools/bench_c2ray_path.py --R 32 --cpu-sources 0 --reps 2 > $R/gpurun_out/pmc_c2ray_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
out=open("$R/gpurun_out/pmc_c2ray_summary.txt","w")
out.write("# rocprofv3 --kernel-trace --pmc <group> -- python3 tools/bench_c2ray_path.py --R 32 --cpu-sources 0 --reps 2\n# kernel counter launches mean-per-launch (FETCH_SIZE / WRITE_SIZE in KiB)\n")
for d in sorted(glob.glob("$R/gpurun_out/pmc_c2ray_[0-9]*/")):
    for f in glob.glob(d+"*/*counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name=r["Kernel_Name"].split("(")[0].replace("void ","").replace("asora::","").split("<")[0]
            if "subbox_sweep" in name or "raytrace_octant" in name: agg[(name,r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k,c),v in sorted(agg.items()):
            line=f"{k:28s} {c:24s} n={len(v)} mean={sum(v)/len(v):.5g}"; print(line); out.write(line+"\n")
PY
rm -rf $R/gpurun_out/pmc_c2ray_[0-9]*
