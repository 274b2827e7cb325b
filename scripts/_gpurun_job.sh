cd $GRAFT_REPO_ROOT
timeout 120 scripts/ubench/mfma_rate 2>&1 | grep "grid  256" | grep "2 exp"
