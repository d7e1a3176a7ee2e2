cd $GRAFT_REPO_ROOT
for v in 4 5 6 8; do
  echo "== SX_SPEC_SIGMAS_CONC=$v"
  STAINX_DIAG=1 SX_SPEC_SIGMAS_CONC=$v timeout -k 10 200 python tools/diag_real.py 2>/dev/null | python3 -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
bad=[r for r in rows if r['slow_slots']]
import statistics
print('tiles',len(rows),'with slow slots',len(bad),[ (r['tile'],r['why']) for r in bad], 'mean cand pct', [round(statistics.mean(r['cand_pct'][s] for r in rows),2) for s in range(4)])
"
done
