cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6ab
AICG_DEV=1 timeout 600 python tools/kbench_w2d_ab.py 12,17 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6ab/kbench_w2d_first_stage_top.txt
