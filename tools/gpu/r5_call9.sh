#!/bin/bash
# round 5, GPU call 9: leaky ReLU in c1 epilogue (c2 without input activation), asm v_max; synth parity at size
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c9; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_g1w.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 600 python tools/kbench_g1w.py -1,0 5 1,3 > $O/kbench_g1w.txt 2>&1; grep -v amdgpu.ids $O/kbench_g1w.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
timeout 900 python -m pytest tests/test_bench_sizes.py tests/test_synth.py -x -q -m gpu -s -k "c1_pipeline or chunk_hubert or synth" > $O/parity.log 2>&1; echo "parity rc=$?"; grep -a "^C1\|synthesizer\|passed\|failed" $O/parity.log | cut -c1-200
python - <<'PY'
import json
for f in ("bench_c3.json",):
    s = json.loads(open("gpurun_out/r5c9/" + f).read().strip().splitlines()[-1])
    print(f, round(s["value"], 1), round(s["ms_per_step"], 1), "stage_s", {k: round(v, 4) for k, v in s["config"]["stage_seconds_per_step"].items()},
          "mdx", round(s["config"]["wall_split_seconds_per_step"]["mdx_s"], 4), "frac", round(s["roofline"]["frac"], 3), "alg", round(s["roofline"]["frac_algorithmic"], 3))
PY
