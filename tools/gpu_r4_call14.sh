#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c14
timeout 300 python tools/kbench_c64_ab.py > gpurun_out/r4c14/c64_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4c14/c64_ab.txt
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4c14/pytest_gpu.log 2>&1
tail -5 gpurun_out/r4c14/pytest_gpu.log
python bench.py --no-cpu-baseline > gpurun_out/r4c14/bench_c3.json 2> gpurun_out/r4c14/bench_c3.err
cut -c1-400 gpurun_out/r4c14/bench_c3.json
python - <<'PY'
import json
s=json.loads(open('gpurun_out/r4c14/bench_c3.json').read().strip().splitlines()[-1])
print(s['value'], s['ms_per_step'])
for st in s['stages']: print(st['stage'], round(st['ms_per_step'],2), round(st['frac'],3))
print({k:v for k,v in s.items() if 'split' in k or 'wall' in k})
PY
