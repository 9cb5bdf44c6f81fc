cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6r
export AICG_DEV=1
timeout 600 python -c "
import aicovergen_amd._lib as L, sys
L._DEFAULT = L._DEFAULT.replace('hip.so', 'hip_dev.so')
import pytest
sys.exit(pytest.main(['tests/test_conv.py', '-q', '-m', 'gpu', '-k', 'winograd_2d and exp15']))" 2>&1 | tail -2
timeout 600 python tools/kbench_w2d_ab.py 12,15,17 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6r/kbench_w2d_half_issue.txt
KB_ONLY=65792 timeout 600 python tools/kbench_w2d_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6r/kbench_w2d_half_issue_phases.txt
