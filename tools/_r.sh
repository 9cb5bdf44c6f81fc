cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6ad
AICG_DEV=1 timeout 600 python tools/kbench_w2d_ab.py 12,17 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6ad/kbench_w2d_fused_epilogue_pairs.txt
