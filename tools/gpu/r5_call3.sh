#!/bin/bash
# round 5, GPU call 3: 1-D Winograd vocoder kernel (conv_g1w): parity, A/B per layer, synthesizer parity at size, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c3; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_g1w.py tests/test_conv_g1s.py tests/test_synth.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 python tools/kbench_g1w.py -1,2,3,4,0 5 > $O/kbench_g1w.txt 2>&1; grep -v amdgpu.ids $O/kbench_g1w.txt
timeout 900 python -m pytest tests/test_bench_sizes.py tests/test_pipeline.py -x -q -m gpu -s > $O/bench_sizes.log 2>&1; echo "bench_sizes rc=$?"; grep -a "^C1\|rel rms\|passed\|failed" $O/bench_sizes.log | cut -c1-220
timeout 600 python bench.py --no-cpu-baseline --conv-shapes $O/conv_shapes_c3.json > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
AICG_WINOGRAD1D=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_c3_no_g1w.json 2> $O/bench_c3_no_g1w.err
python - <<'PY'
import json
for f in ("bench_c3.json", "bench_c3_no_g1w.json"):
    s = json.loads(open("gpurun_out/r5c3/" + f).read().strip().splitlines()[-1])
    print(f, round(s["value"], 1), round(s["ms_per_step"], 1), "stage_s", {k: round(v, 4) for k, v in s["config"]["stage_seconds_per_step"].items()},
          "mdx", round(s["config"]["wall_split_seconds_per_step"]["mdx_s"], 4), "frac", round(s["roofline"]["frac"], 3), "alg", round(s["roofline"]["frac_algorithmic"], 3))
PY
