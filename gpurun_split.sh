export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_split.py tests/test_kernels_misc.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/kbench_tdf.py 2>&1 | tail -7
AICG_PRECISION=bf16x3 timeout 300 python tools/kbench_tdf.py 2>&1 | tail -7
AICG_PRECISION=bf16x3 timeout 120 python tools/kbench_one.py 48 48 3 1 4000000 2>&1 | tail -1
timeout 120 python tools/kbench_one.py 48 48 3 1 4000000 2>&1 | tail -1
AICG_PRECISION=bf16x3 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/split_bench.err | tee gpurun_out/split_bench3.json | cut -c1-300
