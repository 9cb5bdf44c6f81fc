#!/bin/bash
mkdir -p gpurun_out/r4c12
cd /root/repo
export PYTHONUNBUFFERED=1
python bench.py > gpurun_out/r4c12/bench_c3_default.json 2> gpurun_out/r4c12/bench_c3_default.err
cut -c1-220 gpurun_out/r4c12/bench_c3_default.json
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4c12/bench_c3_steps5.json 2>/dev/null
cut -c1-220 gpurun_out/r4c12/bench_c3_steps5.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
