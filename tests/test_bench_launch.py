"""`python bench.py --gpus N` must start N ranks BY ITSELF (the driver's N = 1 command form with N > 1 used to run
one rank and print n_gpus 1) and refuse to print a line for a job size other than the one asked for.  Runs bench.py's real launch
/ sharding / reporting code on the kernel emulator (AICG_BENCH_EMU=1: host tensors, miniature networks, gloo) -- no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=500):
    env = dict(os.environ, AICG_BENCH_EMU="1", AICG_EMU_THREADS="2", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                           "--track-seconds", "0.4"] + extra, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(600)
def test_plain_invocation_with_gpus_2_launches_two_ranks():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                     # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 1 and d["warmup"] == 0
    assert d["config"]["audio_seconds_total"] == pytest.approx(0.8)          # weak scaling: the track grows with N
    per_rank = d["config"]["per_rank_seconds_per_step"]
    assert len(per_rank) == 2 and d["ms_per_step"] == pytest.approx(max(per_rank) * 1e3)   # MAX over ranks
    assert d["value"] == pytest.approx(0.8 / max(per_rank))
    assert d["config"]["workload"].startswith("TEST HOOK")                  # an emulator line can never pass for a measurement


@pytest.mark.timeout(600)
def test_single_rank_line_and_world_size_mismatch():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert d["n_gpus"] == 1 and len(d["config"]["per_rank_seconds_per_step"]) == 1
    # a launcher that started a different number of ranks than --gpus asks for: no line, non-zero exit
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not r.stdout.strip() and "refusing" in r.stderr
