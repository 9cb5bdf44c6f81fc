#!/bin/bash
mkdir -p gpurun_out/r4c7
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_w2d.py > gpurun_out/r4c7/kbench_w2d.log 2>&1
for lvl in "48 256 3072" "96 128 1536" "144 64 768"; do
  tag=$(echo $lvl | cut -d' ' -f1)
  for bits in 0 256 128 1 2 6; do
    AICG_CONV_ABLATE=$bits ABL_NAME="bits $bits" AICG_W2D_WAVES=8 timeout 120 python tools/kbench_w2d_ablate.py $lvl >> gpurun_out/r4c7/ablate_$tag.log 2>&1
  done
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4c7/bench_c3.json 2> gpurun_out/r4c7/bench_c3.err
tail -n 6 gpurun_out/r4c7/kbench_w2d.log
grep -v amdgpu gpurun_out/r4c7/ablate_48.log; grep -v amdgpu gpurun_out/r4c7/ablate_96.log; grep -v amdgpu gpurun_out/r4c7/ablate_144.log
cut -c1-250 gpurun_out/r4c7/bench_c3.json
