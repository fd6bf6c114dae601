cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -m gpu -k "gemm_tn or gemm_nt or swiglu" 2>&1 | tail -3
IADR1_GEMM_TN=2 timeout 600 python tools/gemm_tn_probe.py 20480 2>&1 | grep -v amdgpu.ids
IADR1_GEMM_TN=2 timeout 600 python tools/gemm_tn_probe.py 12288 2>&1 | grep -v amdgpu.ids
