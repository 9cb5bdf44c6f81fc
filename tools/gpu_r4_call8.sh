#!/bin/bash
# round 4, GPU call 8: full gpu test suite, MDX batch sweep, rocprof passes + the round's bench lines
mkdir -p gpurun_out/r4c8
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r4c8/pytest_gpu.log 2>&1
tail -n 4 gpurun_out/r4c8/pytest_gpu.log
for b in 8 11 22; do
  AICG_MDX_BATCH=$b timeout 300 python bench.py --config C2 --steps 4 --warmup 1 --no-cpu-baseline --no-profile-step > gpurun_out/r4c8/bench_c2_batch$b.json 2>/dev/null
  echo "batch $b: $(cut -c1-160 gpurun_out/r4c8/bench_c2_batch$b.json)"
done
bash tools/profile_bench.sh r04 > gpurun_out/r4c8/profile.log 2>&1
tail -n 30 gpurun_out/r4c8/profile.log
