cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_deform_conv.py tests/test_redzone.py tests/test_mxnet_plugin.py -q -m gpu 2>&1 | tail -4
