cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3_pytest1.log)
timeout 300 python tools/c1_triangulate.py --dump gpurun_out/r3_c1_frozen.npz > gpurun_out/r3_c1_frozen.json 2> gpurun_out/r3_c1_frozen.err
AICG_FROZEN_NARROW=0 timeout 300 python tools/c1_triangulate.py --dump gpurun_out/r3_c1_free.npz > gpurun_out/r3_c1_free.json 2> gpurun_out/r3_c1_free.err
timeout 600 python bench.py > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err
timeout 600 python bench.py --preset fp32 --no-cpu-baseline > gpurun_out/r3_bench_fp32preset.json 2> gpurun_out/r3_bench_fp32preset.err
tail -3 gpurun_out/r3_pytest1.log; cat gpurun_out/r3_c1_frozen.json gpurun_out/r3_c1_free.json; head -c 600 gpurun_out/r3_bench_a.json
