#!/bin/bash
mkdir -p gpurun_out/r4c9
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 600 python tools/kbench_w2d_ab.py 2,6,7,8,4,1 > gpurun_out/r4c9/kbench_ab.log 2>&1
grep -v amdgpu gpurun_out/r4c9/kbench_ab.log
