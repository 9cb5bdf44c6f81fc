#!/bin/bash
# round 4, GPU call 2: 2-D Winograd variants + ablation sweeps + full gpu test suite + bench
mkdir -p gpurun_out/r4c2
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_w2d.py > gpurun_out/r4c2/kbench_w2d.log 2>&1
for q in 0 1; do
  AICG_W2D_QUADS=$q timeout 400 python tools/kbench_w2d_ablate.py 48 256 3072 8 > gpurun_out/r4c2/ablate_L0_q$q.log 2>&1
  AICG_W2D_QUADS=$q timeout 400 python tools/kbench_w2d_ablate.py 144 64 768 8 > gpurun_out/r4c2/ablate_L2_q$q.log 2>&1
done
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r4c2/pytest_gpu.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r4c2/bench_c3.json 2> gpurun_out/r4c2/bench_c3.err
AICG_W2D_QUADS=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4c2/bench_c3_q.json 2> gpurun_out/r4c2/bench_c3_q.err
tail -n 6 gpurun_out/r4c2/kbench_w2d.log
tail -n 3 gpurun_out/r4c2/pytest_gpu.log
cut -c1-300 gpurun_out/r4c2/bench_c3.json; cut -c1-300 gpurun_out/r4c2/bench_c3_q.json
