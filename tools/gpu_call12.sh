cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r3_smoke.log
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest12.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3_pytest12.log)
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3_bench_g.json 2> gpurun_out/r3_bench_g.err
AICG_OVERLAP_SYNTH=0 timeout 600 python bench.py --no-cpu-baseline --no-profile-step > gpurun_out/r3_bench_g_nosynthoverlap.json 2> gpurun_out/r3_bench_g2.err
timeout 600 python bench.py --no-cpu-baseline --dump gpurun_out/r3_dump_a.npz --no-profile-step > /dev/null 2>&1
AICG_OVERLAP_SYNTH=0 timeout 600 python bench.py --no-cpu-baseline --dump gpurun_out/r3_dump_b.npz --no-profile-step > /dev/null 2>&1
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/r3_dump_a.npz"); b=np.load("gpurun_out/r3_dump_b.npz")
print("two-stream vs one-stream output equal:", np.array_equal(a["out"], b["out"]), np.array_equal(a["sep"], b["sep"]))
PY
tail -2 gpurun_out/r3_smoke.log; tail -3 gpurun_out/r3_pytest12.log; head -c 300 gpurun_out/r3_bench_g.json; echo; head -c 300 gpurun_out/r3_bench_g_nosynthoverlap.json
