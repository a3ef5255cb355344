# GPU run 9: pipelined copies of the drop-in call
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2i; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipelined_copies or raytrace_matches_oracle" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 300 python tools/pcie_inclusive.py > $O/pcie_inclusive.json 2> $O/pcie.err; cat $O/pcie_inclusive.json; tail -3 $O/pcie.err
timeout -k 10 600 python tools/paper_benchmark.py > $O/paper_protocol.json 2> $O/paper.err; tail -30 $O/paper.err
