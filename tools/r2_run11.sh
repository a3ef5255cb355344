export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:?}; O=$R/gpurun_out/r2k; mkdir -p $O; cd $R
for K in 4 8 12 16; do echo "slabs $K"; ASORA_PIPELINE_SLABS=$K timeout -k 10 300 python tools/pcie_inclusive.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['pipelined_copies']['s_per_call'], d['copies_in_turn']['s_per_call'])"; done
