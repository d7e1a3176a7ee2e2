# sweep of the speculation's knobs over the 150 real crops (diagnostic build): bash tools/diag_real_batches.sh
cd $GRAFT_REPO_ROOT
for kw in 0.10 0.15 0.20 0.30; do for rot in 0.30 0.60; do
  echo "== SX_SPEC_SIGMAS_CONC=8 SX_SPEC_KW=$kw SX_SPEC_ROT=$rot"
  STAINX_DIAG=1 SX_SPEC_SIGMAS_CONC=8 SX_SPEC_KW=$kw SX_SPEC_ROT=$rot timeout -k 10 200 python tools/diag_real_batches.py 2>/dev/null | tail -2
done; done
