cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3_pytest2.log)
timeout 600 python bench.py --no-cpu-baseline --conv-shapes gpurun_out/r3_conv_shapes_c3.json > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err
timeout 600 python bench.py --no-cpu-baseline --config C4 --steps 1 --conv-shapes gpurun_out/r3_conv_shapes_c4.json > gpurun_out/r3_bench_c4.json 2> gpurun_out/r3_bench_c4.err
tail -3 gpurun_out/r3_pytest2.log; head -c 400 gpurun_out/r3_bench_b.json; echo; head -c 400 gpurun_out/r3_bench_c4.json
