export TMPDIR=/tmp
timeout 300 python tools/kbench_gru.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_hubert_rmvpe.py -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_bench_sizes.py -q -m gpu -k rmvpe -s 2>&1 | grep -v "^  frame" | tail -6
