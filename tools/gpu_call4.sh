cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3_pytest4.log)
timeout 600 python bench.py --no-cpu-baseline --conv-shapes gpurun_out/r3_conv_shapes_c3c.json > gpurun_out/r3_bench_d.json 2> gpurun_out/r3_bench_d.err
timeout 600 python bench.py --no-cpu-baseline --config C4 --steps 1 --conv-shapes gpurun_out/r3_conv_shapes_c4c.json > gpurun_out/r3_bench_c4c.json 2> gpurun_out/r3_bench_c4c.err
timeout 600 python bench.py --no-cpu-baseline --preset fp32 > gpurun_out/r3_bench_fp32preset_b.json 2> gpurun_out/r3_bench_fp32preset_b.err
tail -3 gpurun_out/r3_pytest4.log; head -c 300 gpurun_out/r3_bench_d.json; echo; head -c 300 gpurun_out/r3_bench_c4c.json; echo; head -c 300 gpurun_out/r3_bench_fp32preset_b.json
