cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_convnet_ops.py tests/test_gpu_gemm_bnred.py tests/test_gpu_conv_bnbwd.py tests/test_gpu_rn50_step.py -q 2>&1 | tail -3
