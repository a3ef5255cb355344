# A/B of library variants built beforehand (no compiler time on the GPU box): bash tools/ab_prebuilt.sh "old se0 se2" 16 32 64
# Variants are build/variants/libasora_<name>.so (see tools/README.md for how they are made); each is selected through
# PYC2RAY_AMD_LIBASORA (pyc2ray_amd/_capi.py) for its runs.
NAMES=$1; shift
for ROUND in 1 2; do
for V in $NAMES; do
  export PYC2RAY_AMD_LIBASORA=$PWD/build/variants/libasora_$V.so     # the production library is never overwritten
  echo "== $V (round $ROUND)"
  bash tools/sweep_R.sh "$@"
done
done
