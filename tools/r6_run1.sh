#!/bin/bash
# round 6, first GPU job: the new DMA runs of conv_w2d on hardware (parity + A/B), the new end-to-end parity surface, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
timeout 600 python -m pytest tests/test_conv.py -q -m gpu -k "winograd" > $O/pytest_winograd.log 2>&1; echo "winograd rc=$?"; tail -3 $O/pytest_winograd.log
timeout 600 python tools/kbench_w2d_ab.py 2,6 > $O/kbench_w2d_dma_runs.txt 2>&1; echo "kbench rc=$?"; cat $O/kbench_w2d_dma_runs.txt
timeout 1500 python -m pytest tests/test_parity_surface.py -q -m gpu -s > $O/pytest_parity_surface.log 2>&1; echo "parity rc=$?"; tail -5 $O/pytest_parity_surface.log
timeout 600 python bench.py > $O/bench_c3_default.json 2> $O/bench_c3_default.err; echo "bench rc=$?"
python - <<'PY'
import json
s = json.loads(open("gpurun_out/r6a/bench_c3_default.json").read().strip().splitlines()[-1])
print("C3", round(s["value"], 1), round(s["ms_per_step"], 1), "stage_s", {k: round(v, 4) for k, v in s["config"]["stage_seconds_per_step"].items()},
      "mdx", round(s["config"]["wall_split_seconds_per_step"]["mdx_s"], 4), "frac", round(s["roofline"]["frac"], 3))
PY
