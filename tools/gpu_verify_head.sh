#!/bin/bash
# GPU box (through gpurun): the whole -m gpu suite, smoke(), the default bench line -- the checks the driver runs at round end
cd $GRAFT_REPO_ROOT
O=gpurun_out/verify; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "bench rc=$?"
python - <<'PY'
import json
s = json.loads(open("gpurun_out/verify/bench_c3_default.json").read().strip().splitlines()[-1])
print("C3", round(s["value"], 1), round(s["ms_per_step"], 1), "stage_s", {k: round(v, 4) for k, v in s["config"]["stage_seconds_per_step"].items()},
      "mdx", round(s["config"]["wall_split_seconds_per_step"]["mdx_s"], 4), "frac", round(s["roofline"]["frac"], 3), "alg", round(s["roofline"]["frac_algorithmic"], 3),
      "traffic", s["roofline"]["traffic"], "cpu", s["cpu_baseline"]["value"])
PY
