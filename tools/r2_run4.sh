# GPU run 4: A/B of ASORA_LATE_LOOKUP and of the register bound, chemistry with fronts, paper protocol
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2d; mkdir -p $O; cd $R
bash tools/ab_macro.sh ASORA_LATE_LOOKUP "0 1" 8 12 16 20 24 32 48 64 > $O/ab_late_lookup.log 2>&1
bash tools/ab_macro.sh ASORA_MIN_WAVES "5 6" 12 16 20 > $O/ab_min_waves.log 2>&1
make -C pyc2ray_amd/csrc > /dev/null 2>&1
timeout -k 10 600 python tools/chem_front.py > $O/chem_front.json 2> $O/chem_front.err; echo "chem_front exit $?"
bash tools/pmc_chem.sh > $O/pmc_chem.log 2>&1; cp gpurun_out/pmc_chem_front.txt $O/ 2>/dev/null
timeout -k 10 600 python tools/paper_benchmark.py > $O/paper_protocol.json 2> $O/paper_protocol.err; echo "paper exit $?"
cat $O/ab_late_lookup.log $O/ab_min_waves.log
