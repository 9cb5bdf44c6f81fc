#!/bin/bash
mkdir -p gpurun_out/r4c10
cd /root/repo
export PYTHONUNBUFFERED=1
AICG_F0_SEGMENTS=16 timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-step > gpurun_out/r4c10/bench_c3_progressive.json 2> gpurun_out/r4c10/bench_c3_progressive.err
echo "progressive N=1: $(cut -c1-200 gpurun_out/r4c10/bench_c3_progressive.json)"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c10/bench_c3_progressive.json")); print(d["config"]["wall_split_seconds_per_step"])
PY
AICG_FORCE_DEVICE=0 AICG_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-step > gpurun_out/r4c10/bench_2ranks_1gpu.json 2> gpurun_out/r4c10/bench_2ranks_1gpu.err
echo "rc $?"; tail -n 5 gpurun_out/r4c10/bench_2ranks_1gpu.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c10/bench_2ranks_1gpu.json")); print(d["value"], d["n_gpus"], d["config"]["per_rank_seconds_per_step"]); print(d["config"]["per_rank_wall_split_seconds_per_step"])
PY
