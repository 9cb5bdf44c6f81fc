cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r3_smoke.log
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest13.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3_pytest13.log)
timeout 600 python bench.py --no-cpu-baseline --conv-shapes gpurun_out/r3_conv_shapes_final.json > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err
tail -2 gpurun_out/r3_smoke.log; tail -3 gpurun_out/r3_pytest13.log; head -c 300 gpurun_out/r3_bench_final.json
