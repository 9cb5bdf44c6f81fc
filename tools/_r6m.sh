cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6w
timeout 900 python -m pytest tests/test_conv.py tests/test_mdx.py -q -m gpu -x 2>&1 | tail -3
AICG_DEV=1 timeout 600 python tools/kbench_w2d_ab.py 12,15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6w/kbench_w2d_packed_epilogue.txt
