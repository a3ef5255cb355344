export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2j; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipelined_copies or raytrace_matches_oracle" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout -k 10 300 python tools/pcie_inclusive.py > $O/pcie_inclusive.json 2> $O/pcie.err; cat $O/pcie_inclusive.json
cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/tr -- python3 $R/tools/pcie_inclusive.py > $O/trace.log 2>&1; ls $O/tr/*/ | head; 
