cd $GRAFT_REPO_ROOT
timeout 120 python tools/dcn_fused_ablate.py 2>&1 | tail -10
