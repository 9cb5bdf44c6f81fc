export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_split_e2e.py -q -m gpu -s 2>&1 | grep -v "gin_channels\|^  frame" | tail -25
timeout 900 python bench.py --steps 2 --warmup 1 --precision bf16x3 2>gpurun_out/split_bench.err | tee gpurun_out/split_bench4.json | cut -c1-300
