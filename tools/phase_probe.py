"""Where the HuBERT || f0 phase of VC.pipeline goes on the bench track: the phase as shipped, each branch alone (AICG_OVERLAP_F0=0),
and the phase with attention / with the HuBERT GEMMs' time removed (upper bounds of what those kernels can still give)."""
import os, sys, time, torch, numpy as np
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench  # noqa: E402
from aicovergen_amd import ops  # noqa: E402
from synthetic.inputs import vocal_like  # noqa: E402
dev = torch.device("cuda:0")
mdxs, vc, hub, net_g = bench.build_models(dev, "C3", 1, tiny=False, preset="fp16")
audio = torch.from_numpy(vocal_like(240.0, 16000, 1234)).to(dev)


def run(label, n=3):
    ts = []
    for _ in range(n + 1):
        times = [0, 0, 0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vc.pipeline(hub, net_g, 0, audio, "x.wav", times, 0, "rmvpe", "", 0.5, 1, 3, 40000, 0, 0.25, "v2", 0.33, 128, noise_seed=1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0, dict(vc.last_profile), list(times)))
    t, prof, times = sorted(ts[1:], key=lambda x: x[0])[len(ts[1:]) // 2]
    print(f"{label:46s} pipeline {t*1e3:7.1f} ms  f0_s (phase) {prof['f0_s']*1e3:6.1f}  chunks {prof['chunks_s']*1e3:6.1f}  times[hubert,f0,synth] = "
          + ", ".join("%.1f" % (x * 1e3) for x in times), flush=True)


run("as shipped")
os.environ["AICG_OVERLAP_F0"] = "0"
run("serial: f0, then per chunk features + synthesis")
os.environ["AICG_OVERLAP_F0"] = "1"
orig_attn = ops.attention
ops.attention = lambda q, k, v, h, **kw: torch.zeros((q.shape[0], q.shape[1]), dtype=torch.float32, device=q.device)
run("attention removed (upper bound)")
ops.attention = orig_attn
orig_gru = ops.gru_bidir
ops.gru_bidir = lambda gi, whh, bhh, hidden, two_workgroups=None: torch.zeros((2 * hidden, gi.shape[1]), dtype=torch.float32, device=gi.device)
run("GRU removed (upper bound)")
ops.gru_bidir = orig_gru
