cd $GRAFT_REPO_ROOT
timeout 120 python tools/dcn_fused_repeat.py 2>&1 | tail -2
timeout 120 python tools/dcn_fused_time.py 2>&1 | tail -8
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
