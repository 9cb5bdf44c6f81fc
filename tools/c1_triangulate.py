"""BASELINE C1 (30 s through VC.pipeline, full-size networks) triangulated: the HIP output against (a) the reference's own fp32 CPU
output (tests/golden/pipeline_c1_30s.npz) and (b) the float64 evaluation of the same pipeline (tests/golden/pipeline_c1_30s_fp64.npz,
made by tests/golden/make_fp64_c1.py), whose distance from (a) is stored in the fixture.  GPU box; prints one JSON line per run and
(with --dump PATH) stores the decimated output + f0 for offline analysis.  A/B over AICG_* switches by re-running."""
import json, os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import conftest
conftest._bind("hip")
from test_pipeline import build, noise_fn_for
from synthetic import weights
from synthetic.inputs import vocal_like


def dist(a, b):
    d = np.abs(a.astype(np.int64) - b.astype(np.int64))
    return {"rel_rms": float(np.sqrt(np.sum(d.astype(np.float64) ** 2) / np.sum(b.astype(np.float64) ** 2))), "max_lsb": int(d.max()),
            "le1": float((d <= 1).mean()), "exact": float((d == 0).mean())}


def f0dist(f0, ref):
    n = min(len(f0), len(ref)); v = (f0[:n] > 0) & (ref[:n] > 0)
    r = f0[:n][v] / ref[:n][v] - 1
    # what the vocoder's source integrates: the running sum of the f0 error (cycles; 10 ms per frame)
    drift = np.cumsum(np.where(v, f0[:n] - ref[:n], 0.0)) * 0.01
    return {"rel_rms": float(np.sqrt(np.mean(r ** 2))), "rel_max": float(np.abs(r).max()), "rel_mean": float(r.mean()),
            "phase_drift_cycles_max": float(np.abs(drift).max()), "voicing_flips": int(np.sum((f0[:n] > 0) != (ref[:n] > 0)))}


gold = np.load("tests/golden/pipeline_c1_30s.npz")
g64 = np.load("tests/golden/pipeline_c1_30s_fp64.npz")
seed, x = int(gold["seed"][0]), tuple(int(v) for v in gold["x"])
nets = weights.full_model_set(seed)
audio = vocal_like(float(gold["seconds"][0]), 16000, seed + 5)
vc, hub, net_g, tgt_sr = build(conftest.Dev("hip"), nets, x)
out = vc.pipeline(hub, net_g, 0, audio, "x.wav", [0, 0, 0], 0, "rmvpe", "", 0.5, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128, noise_fn=noise_fn_for(nets))
_, audio_pad, opt_ts, p_len = vc.plan(audio)
coarse, f0 = vc.get_f0("x.wav", audio_pad, p_len, 0, "rmvpe", 3, 128)
dec = int(g64["decim"][0])
n = min(len(coarse), len(gold["coarse"]))
res = {"switches": {k: v for k, v in os.environ.items() if k.startswith("AICG_")},
       "hip_vs_reference": dist(out, gold["audio"]), "hip_vs_fp64": dist(out[::dec], g64["audio"]),
       "reference_vs_fp64": {"rel_rms": float(g64["ref_rel_rms"][0]), "max_lsb": int(g64["ref_max_lsb"][0]), "le1": float(g64["ref_le1"][0]),
                             "exact": float(g64["ref_exact"][0])},
       "f0_hip_vs_reference": f0dist(f0, gold["f0"]), "f0_hip_vs_fp64": f0dist(f0, g64["f0"]), "f0_reference_vs_fp64": f0dist(gold["f0"], g64["f0"]),
       "coarse_bins_differ": {"vs_reference": int((coarse[:n] != gold["coarse"][:n]).sum()), "vs_fp64": int((coarse[:n] != g64["coarse"][:n]).sum())}}
print(json.dumps(res))
if "--dump" in sys.argv:
    np.savez_compressed(sys.argv[sys.argv.index("--dump") + 1], audio=out[::dec], f0=f0, coarse=coarse)
