#!/bin/bash
# round 5, GPU call 6: conv_g1w with dilations 3 / 5: parity, A/B per layer, synthesizer + C1 parity at size, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c6; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_g1w.py tests/test_synth.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 600 python tools/kbench_g1w.py -1,0 5 3,5 > $O/kbench_g1w_dilated.txt 2>&1; grep -v amdgpu.ids $O/kbench_g1w_dilated.txt
timeout 900 python -m pytest tests/test_bench_sizes.py -x -q -m gpu -s -k "c1_pipeline or chunk_hubert" > $O/bench_sizes.log 2>&1; echo "bench_sizes rc=$?"; grep -a "^C1\|rel rms\|passed\|failed" $O/bench_sizes.log | cut -c1-220
timeout 600 python bench.py --no-cpu-baseline --conv-shapes $O/conv_shapes_c3.json > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
python - <<'PY'
import json
s = json.loads(open("gpurun_out/r5c6/bench_c3.json").read().strip().splitlines()[-1])
print("C3", round(s["value"], 1), round(s["ms_per_step"], 1), "stage_s", {k: round(v, 4) for k, v in s["config"]["stage_seconds_per_step"].items()},
      "mdx", round(s["config"]["wall_split_seconds_per_step"]["mdx_s"], 4), "frac", round(s["roofline"]["frac"], 3), "alg", round(s["roofline"]["frac_algorithmic"], 3))
PY
