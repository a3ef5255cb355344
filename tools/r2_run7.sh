# GPU run 7: PCIe microbenchmark, largest-first unit order A/B, chemistry register-bound A/B
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2g; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/micro/pcie.hip -o /tmp/pcie && timeout -k 10 120 /tmp/pcie > $O/pcie.txt 2>&1; cat $O/pcie.txt
bash tools/ab_macro.sh ASORA_UNITS_LARGEST_FIRST "0 1" 24 32 48 64 > $O/ab_units_order.log 2>&1; cat $O/ab_units_order.log
bash tools/ab_macro.sh ASORA_CHEM_MIN_WAVES "1 8" 32 > $O/ab_chem_waves.log 2>&1; cat $O/ab_chem_waves.log
make -C pyc2ray_amd/csrc > /dev/null 2>&1
timeout -k 10 300 python tools/test1_stromgren.py --cpu-steps 0 > $O/test1.json 2> $O/test1.err; cat $O/test1.json
