cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6y
for v in 0 1 0 1; do
AICG_DEV=1 AICG_RB_STREAMS=$v timeout 600 python bench.py --no-cpu-baseline --no-profile-step --steps 3 > gpurun_out/r6y/b$v.json 2>/dev/null
python - $v <<'PY'
import json,sys
s=json.loads(open("gpurun_out/r6y/b%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("RB_STREAMS", sys.argv[1], round(s["ms_per_step"],1), s["config"]["stage_seconds_per_step"])
PY
done
