# The raytrace on an EVOLVING field (tools/chem_front.py: configs[3] with fluxes x 1e3, second time step) with parts switched off in
# the prebuilt diagnostic library (build/variants/libasora_abl.so; wrong results by design): what the incoherent table lookups, the
# atomics and the nHI loads cost there.  usage (GPU box): bash tools/ablate_front.sh "0 1 8 9 128"
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it) to the repository root}" || exit 1
export PYC2RAY_AMD_LIBASORA=$PWD/build/variants/libasora_abl.so
for A in $1; do
  ASORA_ABLATE=$A timeout -k 10 300 python tools/chem_front.py --histogram 0 2>/dev/null | tail -1 > gpurun_out/abl_front.json
  python -c "
import json;d=json.load(open('gpurun_out/abl_front.json'))
q=d['quiet']['step_2']; f=d['fronts']['step_2']
print('ablate $A quiet raytrace ms', [round(v,3) for v in q['raytrace_ms']], 'fronts raytrace ms', [round(v,3) for v in f['raytrace_ms']][:6], 'iters', f['outer_iterations'])"
done
