# A/B of a compile-time macro of the HIP library on ONE box: bash tools/ab_macro.sh ASORA_LATE_ATOMIC "0 1" 16 32 64
# (rebuilds the library per value, restores the default build at the end)
# whatever happens, leave the default build behind (a diagnostic build gives WRONG results under ASORA_ABLATE)
trap 'make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc > /dev/null 2>&1' EXIT
M=$1; VALS=$2; shift 2
for V in $VALS; do
  make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc EXTRA=-D$M=$V > /dev/null 2>&1
  echo "== $M=$V"
  bash tools/sweep_R.sh "$@"
done
make -C pyc2ray_amd/csrc clean > /dev/null; make -C pyc2ray_amd/csrc > /dev/null 2>&1
