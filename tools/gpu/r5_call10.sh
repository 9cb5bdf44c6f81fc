#!/bin/bash
# round 5, GPU call 10: conv_g1w on the short maps of enc_p / flow too (k = 3 FFN layers, k = 5 WaveNet layers at ~6 500 positions)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c10; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_g1w.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for m in 16384 4096; do
  AICG_WINOGRAD1D_MIN=$m timeout 600 python bench.py --no-cpu-baseline --conv-shapes $O/conv_shapes_min$m.json > $O/bench_c3_min$m.json 2> $O/bench_c3_min$m.err
done
AICG_WINOGRAD1D_MIN=4096 timeout 900 python -m pytest tests/test_bench_sizes.py tests/test_synth.py tests/test_pipeline.py -x -q -m gpu -s -k "c1_pipeline or chunk_hubert or synth or pipeline" > $O/parity_min4096.log 2>&1; echo "parity rc=$?"; grep -a "^C1\|synthesizer\|passed\|failed" $O/parity_min4096.log | cut -c1-200
python - <<'PY'
import json
for m in (16384, 4096):
    s = json.loads(open("gpurun_out/r5c10/bench_c3_min%d.json" % m).read().strip().splitlines()[-1])
    print(m, round(s["value"], 1), round(s["ms_per_step"], 1), "stage_s", {k: round(v, 4) for k, v in s["config"]["stage_seconds_per_step"].items()},
          "mdx", round(s["config"]["wall_split_seconds_per_step"]["mdx_s"], 4), "frac", round(s["roofline"]["frac"], 3))
    rows = json.load(open("gpurun_out/r5c10/conv_shapes_min%d.json" % m))
    for r in rows:
        if (" k1x5 " in r["shape"] or ("k1x3" in r["shape"] and ("C192>768" in r["shape"] or "C768>192" in r["shape"]))) and "6420" in r["shape"]:
            print("   ", r["shape"], r["launches"], round(r["ms"], 3), round(r["tflops"], 1))
PY
