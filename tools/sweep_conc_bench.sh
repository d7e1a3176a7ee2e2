cd $GRAFT_REPO_ROOT
for v in 5 8 10; do
  echo "== SX_SPEC_SIGMAS_CONC=$v"
  STAINX_DIAG=1 SX_SPEC_SIGMAS_CONC=$v timeout -k 10 300 python bench.py --no-cpu --steps 300 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['real_tiles']
print('synthetic', d['ms_per_step'], 'real', r['ms_per_step'], 'median', r['device_ms_median'], 'slow slots last call', r['slow_slots_last_call'], r['first_calls_ms'])
"
done
