# A/B of prebuilt library variants on the EVOLVING field (tools/chem_front.py): bash tools/ab_front_prebuilt.sh "nt0 nt1 nt2"
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
for ROUND in 1 2; do for V in $1; do
  export PYC2RAY_AMD_LIBASORA=$PWD/build/variants/libasora_$V.so
  timeout -k 10 300 python tools/chem_front.py --histogram 0 2>/dev/null | tail -1 > gpurun_out/abf.json
  python -c "
import json,numpy as np;d=json.load(open('gpurun_out/abf.json'))
q=d['quiet']['step_2']; f=d['fronts']['step_2']
print('$V round $ROUND: quiet raytrace', round(float(np.mean(q['raytrace_ms'])),4), 'pass', round(float(np.mean(q['chemistry_ms'])),4), '| fronts raytrace mean', round(float(np.mean(f['raytrace_ms'])),4), 'pass', round(float(np.mean(f['chemistry_ms'])),4), 'iters', f['outer_iterations'])"
done; done
