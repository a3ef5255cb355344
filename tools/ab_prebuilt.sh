# A/B of library variants built beforehand (no compiler time on the GPU box): bash tools/ab_prebuilt.sh "old se0 se2" 16 32 64
# Variants are build/variants/libasora_<name>.so (see tools/README.md for how they are made); each is copied over the
# library of THIS checkout for its runs, and the default build is put back at the end.
cp pyc2ray_amd/lib/libasora_hip.so build/variants/libasora_default_saved.so
trap 'cp build/variants/libasora_default_saved.so pyc2ray_amd/lib/libasora_hip.so' EXIT
NAMES=$1; shift
for ROUND in 1 2; do
for V in $NAMES; do
  cp build/variants/libasora_$V.so pyc2ray_amd/lib/libasora_hip.so
  echo "== $V (round $ROUND)"
  bash tools/sweep_R.sh "$@"
done
done
