cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest7.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3_pytest7.log)
timeout 600 python bench.py --no-cpu-baseline --conv-shapes gpurun_out/r3_conv_shapes_c3e.json > gpurun_out/r3_bench_f.json 2> gpurun_out/r3_bench_f.err
timeout 600 python bench.py --no-cpu-baseline --config C2 > gpurun_out/r3_bench_c2b.json 2> gpurun_out/r3_bench_c2b.err
tail -3 gpurun_out/r3_pytest7.log; head -c 300 gpurun_out/r3_bench_f.json; echo; head -c 300 gpurun_out/r3_bench_c2b.json
