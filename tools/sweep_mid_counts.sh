cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
for NS in 64 125 250; do for S in 0 2 3; do for T in 0 64 128 256 512; do
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-sources 0 --nsrc $NS --sectors $S --block-threads $T > /tmp/o.json 2>/dev/null && python -c "
import json;d=json.load(open('/tmp/o.json'));print('nsrc $NS sectors $S threads $T raytrace', round(d['kernels_ms_per_step']['raytrace'],4))"
done; done; done
