export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv.py tests/test_conv_fuzz.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1; do
AICG_CONV_V3_160=$v timeout 120 python tools/kbench_2d_one.py 144 144 3 16 64 768 2>&1 | tail -1
AICG_CONV_V3_160=$v timeout 120 python tools/kbench_one.py 160 160 7 1 600000 2>&1 | tail -1
AICG_CONV_V3_160=$v timeout 120 python tools/kbench_one.py 480 480 3 1 200000 2>&1 | tail -1
done
for v in 0 1; do
AICG_CONV_V3_160=$v timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v3_160=$v', round(d['value'],1), round(d['roofline']['achieved'],1), d['config']['wall_split_seconds_per_step']['mdx_s'])"
done
