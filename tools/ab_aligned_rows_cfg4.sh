# A/B of ASORA_OPT_ALIGNED_ROWS on BASELINE configs[4] on one GPU (512^3, 1e5 sources): bash tools/ab_aligned_rows_cfg4.sh
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
mkdir -p gpurun_out/r3z
for round in 1 2; do
  for mode in off auto; do
    if [ $mode = off ]; then export PYC2RAY_AMD_OPTIONS="15=1"; else unset PYC2RAY_AMD_OPTIONS; fi
    timeout -k 10 400 python bench.py --N 512 --nsrc 100000 --workload cosmo --steps 3 --warmup 1 --repeats 3 --cpu-sources 0 2>/dev/null > gpurun_out/abc.json || exit 1
    python -c "
import json;d=json.load(open('gpurun_out/abc.json'));k=d['kernels_ms_per_step']
print('cfg4 aligned rows $mode: step ms %.3f raytrace %.3f chemistry %.3f value %.4e' % (d['ms_per_step'], k['raytrace'], k['chemistry'], d['value']))"
  done
done
