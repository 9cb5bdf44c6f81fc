#!/bin/bash
# round 5, GPU call 1: RCCL on one rank, C1 under both schedules, f0 bias analysis, this round's starting bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c1; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_rccl_one_rank.py -x -q -m gpu -s > $O/rccl_test.log 2>&1; echo "rccl test rc=$?"; tail -3 $O/rccl_test.log
cp gpurun_out/r05_rccl_one_rank.json $O/ 2>/dev/null
timeout 900 python -m pytest tests/test_bench_sizes.py -x -q -m gpu -s -k "c1_pipeline" > $O/c1_tests.log 2>&1; echo "c1 tests rc=$?"; grep -a "^C1\|passed\|failed" $O/c1_tests.log
timeout 600 python tools/c1_f0_bias.py --out $O/r05_c1_f0_bias.json --tracks $O/r05_c1_f0_tracks.npz > $O/bias.log 2>&1; echo "bias rc=$?"; tail -1 $O/bias.log | cut -c1-1500
timeout 600 python -m pytest tests/test_pipeline.py tests/test_conv_g1k.py -x -q -m gpu > $O/pipeline_tests.log 2>&1; echo "pipeline tests rc=$?"; tail -2 $O/pipeline_tests.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
AICG_DIST_BACKEND=nccl AICG_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c3_one_rank_rccl.json 2> $O/bench_c3_one_rank_rccl.err; echo "bench rccl rc=$?"
python - <<'PY'
import json
for f in ("bench_c3.json", "bench_c3_one_rank_rccl.json"):
    try:
        s = json.loads(open("gpurun_out/r5c1/" + f).read().strip().splitlines()[-1])
        print(f, round(s["value"], 1), round(s["ms_per_step"], 1), s["config"].get("collectives"), s["config"].get("per_rank_wall_split_seconds_per_step"))
        print("  stage_s", s["config"]["stage_seconds_per_step"], "split", {k: round(v, 4) for k, v in s["config"]["wall_split_seconds_per_step"].items()})
        if s.get("roofline"): print("  roofline frac", round(s["roofline"]["frac"], 3), "alg", round(s["roofline"]["frac_algorithmic"], 3))
    except Exception as e:
        print(f, "unreadable:", e)
PY
