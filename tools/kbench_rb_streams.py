"""The vocoder's ResBlock chains on their own streams (AICG_RB_STREAMS=1, opt-in) against the one-stream walk, round-robin on the bench track's
RVC stage (VC.pipeline: HuBERT || f0 phase, then per chunk front + vocoder) and on one reference-sized synthesizer chunk."""
import os, sys, time, statistics, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from synthetic.inputs import vocal_like  # noqa: E402
dev = torch.device("cuda:0")
mdxs, vc, hub, net_g = bench.build_models(dev, "C3", 1, tiny=False, preset="fp16")
audio = torch.from_numpy(vocal_like(240.0, 16000, 1234)).to(dev)


def once():
    times = [0, 0, 0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vc.pipeline(hub, net_g, 0, audio, "x.wav", times, 0, "rmvpe", "", 0.5, 1, 3, 40000, 0, 0.25, "v2", 0.33, 128, noise_seed=1)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, vc.last_profile["chunks_s"]


res = {"0": [], "1": []}
for mode in ("0", "1"):
    os.environ["AICG_RB_STREAMS"] = mode
    once()
for r in range(6):
    for mode in (("0", "1") if r % 2 == 0 else ("1", "0")):
        os.environ["AICG_RB_STREAMS"] = mode
        res[mode].append(once())
for mode in ("0", "1"):
    print(f"AICG_RB_STREAMS={mode}: pipeline {statistics.median(t for t, _ in res[mode])*1e3:7.1f} ms   chunk loop {statistics.median(c for _, c in res[mode])*1e3:7.1f} ms"
          f"   (min {min(t for t, _ in res[mode])*1e3:7.1f})", flush=True)
