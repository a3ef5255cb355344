# One source, whole-box trace (BASELINE configs[1]-like): raytrace time per launch over decompositions and workgroup sizes
# usage: bash tools/sweep_single_source.sh [N] [R]
N=${1:-128}; RR=${2:-1000}
for S in 2 4; do for T in 256 512 1024; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-sources 0 --N $N --nsrc 1 --R $RR --sectors $S --block-threads $T > /tmp/ss.json 2>/dev/null && python -c "
import json;d=json.load(open('/tmp/ss.json'));print('N $N R $RR sectors $S threads $T raytrace ms', round(d['kernels_ms_per_step']['raytrace'],4))"
done; done
timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-sources 0 --N $N --nsrc 1 --R $RR > /tmp/ss.json 2>/dev/null && python -c "
import json;d=json.load(open('/tmp/ss.json'));print('N $N R $RR auto raytrace ms', round(d['kernels_ms_per_step']['raytrace'],4))"
