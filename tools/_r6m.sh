cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6q
export AICG_DEV=1
KB_ONLY=256,262400 timeout 600 python tools/kbench_w2d_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6q/kbench_w2d_phases_by_wave.txt
timeout 600 python tools/kbench_w2d_ab.py 12,18,19 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6q/kbench_w2d_prio.txt
