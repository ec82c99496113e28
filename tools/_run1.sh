cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_deform_conv.py -q -m gpu -k "fused" 2>&1 | tail -5
timeout 120 python tools/dcn_fused_repeat.py 2>&1 | tail -3
timeout 120 python tools/dcn_fused_time.py 2>&1 | tail -8
timeout 120 python tools/dcn_fused_ablate.py 2>&1 | tail -8
timeout 120 python tools/dcn_fused_clocks.py 2>&1 | tail -4
