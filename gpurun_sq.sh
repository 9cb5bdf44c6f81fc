export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_split.py tests/test_conv.py -x -q -m gpu 2>&1 | tail -2
export AICG_PRECISION=bf16x3
timeout 120 python tools/kbench_one.py 256 256 7 1 400000 2>&1 | tail -1
timeout 120 python tools/kbench_one.py 128 128 11 5 800000 2>&1 | tail -1
timeout 120 python tools/kbench_one.py 64 64 11 5 1600000 2>&1 | tail -1
timeout 120 python tools/kbench_2d_one.py 48 48 3 16 256 3072 2>&1 | tail -1
timeout 120 python tools/kbench_2d_one.py 144 144 3 16 64 768 2>&1 | tail -1
timeout 120 python tools/kbench_2d_one.py 240 240 3 16 16 192 2>&1 | tail -1
unset AICG_PRECISION
timeout 120 python tools/kbench_one.py 128 128 7 3 800000 2>&1 | tail -1
for i in 1 2; do
timeout 900 python bench.py --precision bf16x3 --no-cpu-baseline 2>/dev/null | cut -c1-160
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-160
done
