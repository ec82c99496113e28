cd $GRAFT_REPO_ROOT
for ab in 0 60 63; do echo "== ablate $ab"; bash tools/kt_ops.sh ktdcn$ab deform_conv dcn_fused_ablate=$ab 2>&1 | grep "gemm_pre\|gemm_prep"; done
