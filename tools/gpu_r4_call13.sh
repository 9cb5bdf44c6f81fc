#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c13
timeout 600 python tools/kbench_attn_ab.py 2 > gpurun_out/r4c13/attn_ab.txt 2>&1
timeout 400 python tools/kbench_gemm_probe.py > gpurun_out/r4c13/gemm_probe.txt 2>&1
cat gpurun_out/r4c13/attn_ab.txt | grep -v amdgpu.ids
cat gpurun_out/r4c13/gemm_probe.txt | grep -v amdgpu.ids
