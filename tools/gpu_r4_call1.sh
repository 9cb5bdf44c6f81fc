#!/bin/bash
# round 4, GPU call 1: 2-D Winograd kernel micro-benchmark + new parity tests + a first bench line
mkdir -p gpurun_out/r4c1
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_w2d.py > gpurun_out/r4c1/kbench_w2d.log 2>&1
echo "kbench rc $?" >> gpurun_out/r4c1/kbench_w2d.log
timeout 600 python -m pytest tests/test_conv.py -q -m gpu -k "winograd_at_mdx" -s > gpurun_out/r4c1/test_wino_levels.log 2>&1
timeout 600 python -m pytest tests/test_crepe.py -q -m gpu -s > gpurun_out/r4c1/test_crepe.log 2>&1
timeout 900 python -m pytest tests/test_bench_sizes.py -q -m gpu -s -k "c1_pipeline" > gpurun_out/r4c1/test_c1.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r4c1/bench_c3.json 2> gpurun_out/r4c1/bench_c3.err
timeout 600 python bench.py --config C2 --steps 5 --warmup 2 > gpurun_out/r4c1/bench_c2.json 2> gpurun_out/r4c1/bench_c2.err
AICG_WINOGRAD=1 timeout 600 python bench.py --config C2 --steps 5 --warmup 2 > gpurun_out/r4c1/bench_c2_rows.json 2> gpurun_out/r4c1/bench_c2_rows.err
tail -n 8 gpurun_out/r4c1/kbench_w2d.log
tail -n 3 gpurun_out/r4c1/test_wino_levels.log gpurun_out/r4c1/test_crepe.log gpurun_out/r4c1/test_c1.log
cat gpurun_out/r4c1/bench_c3.json | cut -c1-600
