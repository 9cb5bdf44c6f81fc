#!/bin/bash
# round 5, GPU call 2: stride-2 extractor kernel (conv_g1s) A/B + tests, input-centric col2im, C1 noise study with independent samples, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c2; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_g1s.py tests/test_kernels_misc.py tests/test_hubert_rmvpe.py tests/test_synth.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 300 python tools/kbench_g1s.py 1,2,3,0 7 > $O/kbench_g1s.txt 2>&1; grep -v amdgpu.ids $O/kbench_g1s.txt
timeout 900 python tools/c1_f0_bias.py --out $O/r05_c1_f0_bias.json --tracks $O/r05_c1_f0_tracks.npz --samples > $O/bias.log 2>&1; echo "bias rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r5c2/r05_c1_f0_bias.json"))
print(json.dumps(r.get("noise_model"), indent=0))
for s in r.get("independent_samples", []): print(s)
PY
timeout 900 python -m pytest tests/test_bench_sizes.py -x -q -m gpu > $O/bench_sizes.log 2>&1; echo "bench_sizes rc=$?"; tail -2 $O/bench_sizes.log
timeout 600 python bench.py --no-cpu-baseline --conv-shapes $O/conv_shapes_c3.json > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
python - <<'PY'
import json
s = json.loads(open("gpurun_out/r5c2/bench_c3.json").read().strip().splitlines()[-1])
print("C3", round(s["value"], 1), round(s["ms_per_step"], 1))
print("  stage_s", s["config"]["stage_seconds_per_step"], "split", {k: round(v, 4) for k, v in s["config"]["wall_split_seconds_per_step"].items()})
print("  roofline frac", round(s["roofline"]["frac"], 3), "alg", round(s["roofline"]["frac_algorithmic"], 3))
for st in s["stages"]: print("  ", st["stage"], round(st["ms_per_step"], 2), round(st["frac"], 3))
PY
