export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv.py tests/test_conv_fuzz.py tests/test_mdx.py tests/test_hubert_rmvpe.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1; do
AICG_CONV_M16H=$v timeout 120 python tools/kbench_2d_one.py 48 48 3 16 256 3072 2>&1 | tail -1
AICG_CONV_M16H=$v timeout 120 python tools/kbench_2d_one.py 16 16 3 1 2048 1024 2>&1 | tail -1
done
for v in 0 1; do
AICG_CONV_M16H=$v timeout 900 python bench.py --config C2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('m16h=$v C2', round(d['value'],1), round(d['roofline']['achieved'],1), d['config']['wall_split_seconds_per_step']['mdx_s'])"
done
