cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/kbench_gru.py > gpurun_out/r3_kbench_gru.log 2>&1
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3_pytest3.log)
timeout 600 python bench.py --no-cpu-baseline --conv-shapes gpurun_out/r3_conv_shapes_c3b.json > gpurun_out/r3_bench_c.json 2> gpurun_out/r3_bench_c.err
timeout 600 python bench.py --no-cpu-baseline --config C4 --steps 1 > gpurun_out/r3_bench_c4b.json 2> gpurun_out/r3_bench_c4b.err
AICG_GRU_WG=2 timeout 600 python bench.py --no-cpu-baseline --no-profile-step > gpurun_out/r3_bench_c_gru2.json 2> gpurun_out/r3_bench_c_gru2.err
cat gpurun_out/r3_kbench_gru.log; tail -3 gpurun_out/r3_pytest3.log; head -c 300 gpurun_out/r3_bench_c.json; echo; head -c 300 gpurun_out/r3_bench_c4b.json; echo; head -c 300 gpurun_out/r3_bench_c_gru2.json
