cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6t
timeout 900 python -m pytest tests/test_half.py tests/test_conv.py -q -m gpu -x 2>&1 | tail -3
AICG_DEV=1 timeout 600 python tools/kbench_w2d_ab.py 12,14 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6t/kbench_w2d_tied_settled.txt
timeout 600 python tools/kbench_half.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6t/kbench_half.txt
timeout 900 python bench.py --precision f16 --no-cpu-baseline > gpurun_out/r6t/bench_c3_f16.json 2> gpurun_out/r6t/bench_c3_f16.err; echo "bench rc=$?"
python - <<'PY'
import json
s = json.loads(open("gpurun_out/r6t/bench_c3_f16.json").read().strip().splitlines()[-1])
print("C3 f16", round(s["value"], 1), round(s["ms_per_step"], 1), s["config"]["stage_seconds_per_step"], s["config"]["wall_split_seconds_per_step"]["mdx_s"])
PY
