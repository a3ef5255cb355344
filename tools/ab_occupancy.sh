cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
# whatever happens, leave the default build behind
trap 'make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc > /dev/null 2>&1' EXIT
for CFG in "-DASORA_LATE_LOOKUP=0" "-DASORA_LATE_LOOKUP=0 -DASORA_MIN_WAVES=5" "-DASORA_LATE_LOOKUP=0" "-DASORA_LATE_LOOKUP=0 -DASORA_MIN_WAVES=5" "-DASORA_LATE_LOOKUP=1"; do
  make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc EXTRA="$CFG" > /dev/null 2>&1
  echo "== $CFG"; bash tools/sweep_R.sh 16 24 32
done
make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc > /dev/null 2>&1
