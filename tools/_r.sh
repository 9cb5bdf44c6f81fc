cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6aa
timeout 600 python tools/kbench_mdx_us.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6aa/kbench_mdx_us.txt
