#!/bin/bash
# round 5, GPU call 7: the C1 noise study over 32 other inputs, the whole -m gpu suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c7; mkdir -p $O
timeout 1200 python tools/c1_f0_bias.py --out $O/r05_c1_f0_bias_32_inputs.json --samples > $O/bias.log 2>&1; echo "bias rc=$?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r5c7/r05_c1_f0_bias_32_inputs.json"))
print(json.dumps(r.get("across_inputs"), indent=0))
for s in r.get("independent_samples", []):
    print({k: (round(v, 10) if isinstance(v, float) else v) for k, v in s.items() if k != "flips"}, (s.get("flips") or [])[:3])
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
