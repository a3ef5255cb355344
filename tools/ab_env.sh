# A/B of an environment switch of the library on ONE box: bash tools/ab_env.sh ASORA_PAIRS_BY_XCD "0 1 0 1" 28 32 40
V=$1; VALS=$2; shift 2
for X in $VALS; do echo "== $V=$X"; env $V=$X bash tools/sweep_R.sh "$@"; done
