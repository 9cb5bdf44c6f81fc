cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/kbench_attn.py > gpurun_out/r3_kbench_attn.log 2>&1
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3_pytest5.log)
timeout 600 python bench.py --no-cpu-baseline --conv-shapes gpurun_out/r3_conv_shapes_c3d.json > gpurun_out/r3_bench_e.json 2> gpurun_out/r3_bench_e.err
timeout 600 python bench.py --no-cpu-baseline --config C2 > gpurun_out/r3_bench_c2.json 2> gpurun_out/r3_bench_c2.err
cat gpurun_out/r3_kbench_attn.log; tail -3 gpurun_out/r3_pytest5.log; head -c 300 gpurun_out/r3_bench_e.json; echo; head -c 300 gpurun_out/r3_bench_c2.json
