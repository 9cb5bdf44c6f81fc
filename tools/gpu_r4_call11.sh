#!/bin/bash
mkdir -p gpurun_out/r4c11
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 600 python tools/phase_probe.py > gpurun_out/r4c11/phase_probe.log 2>&1
grep -v "amdgpu\|gin_channels" gpurun_out/r4c11/phase_probe.log | tail -8
