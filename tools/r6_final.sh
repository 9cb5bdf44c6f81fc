#!/bin/bash
# round 6, final GPU verification at HEAD: the whole -m gpu suite (with the parity surface's numbers), smoke(), the default bench line,
# and the same line with every join forced through a one-rank RCCL group (the collective path's overhead at the same commit)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_final; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -s > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench_c3_10_steps.json 2> $O/bench_c3_10_steps.err; echo "bench rc=$?"
AICG_DIST_BACKEND=nccl AICG_FORCE_COLLECTIVES=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_one_rank_rccl.json 2> $O/bench_c3_one_rank_rccl.err; echo "rccl bench rc=$?"
python - <<'PY'
import json
for f in ("bench_c3_10_steps", "bench_c3_one_rank_rccl"):
    s = json.loads(open("gpurun_out/r6_final/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(s["value"], 1), round(s["ms_per_step"], 1), "stage_s", {k: round(v, 4) for k, v in s["config"]["stage_seconds_per_step"].items()},
          "mdx", round(s["config"]["wall_split_seconds_per_step"]["mdx_s"], 4), "frac", round(s["roofline"]["frac"], 3), s["config"]["collectives"],
          s["config"]["per_rank_wall_split_seconds_per_step"])
PY
