"""Is the HIP f0 error on BASELINE C1 a BIAS or noise?  (VERDICT r4 weak #1 / next #3b.)

Free-running C1 sits 4.5-5.5e-4 relative RMS from the reference's waveform while the reference sits 1.0-1.8e-4 from itself (other host)
and from float64 -- with f0 tracks that are equally accurate (2.3-2.6e-7 relative RMS against float64).  The vocoder's harmonic source
integrates f0 (SineGen: phase[n] = sum f0 / sr, reference src/infer_pack/models.py:320-370), so what reaches the waveform is the RUNNING
SUM of the f0 error, and a bias would integrate linearly where noise grows as sqrt(t).  This tool measures exactly that, for three
tracks over the same 3 601 frames:

    HIP (this run, default progressive schedule)   |   the reference's own fp32 CPU track (tests/golden/pipeline_c1_30s.npz)
    float64 evaluation (tests/golden/pipeline_c1_30s_fp64.npz) = stand-in for exact arithmetic

per pair: signed mean of the relative error and its standard error (lag-1-autocorrelation-corrected effective sample size), the running
sum in cycles (10 ms per frame) -- end of chunk, largest excursion, and the excursion an unbiased random walk with the measured
per-frame variance and autocorrelation would make --, and how the waveform distance follows the accumulated phase difference over time
(per 100 ms window: relative waveform error against |phase difference|).  GPU box:

    python tools/c1_f0_bias.py --out profiles/r05_c1_f0_bias.json [--tracks profiles/r05_c1_f0_tracks.npz] [--samples]

--samples adds INDEPENDENT samples of the same quantity (how far two equally accurate fp32 evaluations land from each other):
the C1 input under other, equally valid HIP summation orders (single-workgroup GRU; one-launch schedule), and eight other inputs through
the same networks against the reference's own run on them (tests/golden/pipeline_c1_30s_audio2001..2008.npz, make_golden.py c1seeds);
and a Monte-Carlo of the noise model -- unbiased white per-frame f0 error of the measured spread, waveform error = slope x |phase
difference| as measured -- that turns the spread into quantiles of the waveform distance: the test's gate is read off it.
"""
import json
import os
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pair_stats(a, b, name):
    """f0 track a against track b (Hz per 10 ms frame): signed statistics of what the source integrates."""
    n = min(len(a), len(b))
    a, b = a[:n], b[:n]
    v = (a > 0) & (b > 0)
    r = a[v] / b[v] - 1                                    # relative error, voiced frames
    dhz = np.where(v, a - b, 0.0)                          # Hz; unvoiced frames contribute no phase
    drift = np.cumsum(dhz) * 0.01                          # cycles of the fundamental (10 ms per frame)
    m, s = float(r.mean()), float(r.std(ddof=1))
    rc = r - m
    rho1 = float((rc[1:] * rc[:-1]).sum() / (rc * rc).sum())
    # autocorrelation time from the first lags that stay positive (Sokal's window would need more frames than a 36 s chunk has)
    acf = [1.0]
    for lag in range(1, 200):
        c = float((rc[lag:] * rc[:-lag]).sum() / (rc * rc).sum())
        if c <= 0.05:
            break
        acf.append(c)
    tau = 1.0 + 2.0 * sum(acf[1:])                         # integrated autocorrelation time (frames)
    n_eff = len(r) / tau
    se = s / np.sqrt(n_eff)
    # an UNBIASED walk with this per-frame spread and correlation: rms end-point = sigma_hz * 0.01 * sqrt(n * tau)
    sig_hz = float(dhz[v].std(ddof=1))
    walk_rms_end = sig_hz * 0.01 * np.sqrt(v.sum() * tau)
    bias_end = float(dhz[v].mean()) * 0.01 * v.sum()       # what the signed mean alone integrates to
    return {"pair": name, "voiced_frames": int(v.sum()), "voicing_flips": int(((a > 0) != (b > 0)).sum()),
            "rel_rms": float(np.sqrt((r ** 2).mean())), "rel_max": float(np.abs(r).max()),
            "rel_mean_signed": m, "rel_mean_standard_error": float(se), "mean_over_standard_error": float(m / se),
            "lag1_autocorrelation": rho1, "autocorrelation_time_frames": float(tau), "effective_samples": float(n_eff),
            "phase_end_of_chunk_cycles": float(drift[-1]), "phase_max_excursion_cycles": float(np.abs(drift).max()),
            "phase_from_signed_mean_alone_cycles": bias_end,
            "phase_rms_endpoint_of_an_unbiased_walk_cycles": float(walk_rms_end),
            "verdict": ("bias" if abs(m / se) > 3 else "consistent with zero-mean correlated noise (|mean| < 3 standard errors)")}, drift


def follow(out, ref, drift, sr_frames=400, win=10):
    """Does the waveform distance follow the accumulated phase difference?  Per window of `win` frames: relative RMS waveform error
    against the mean |phase difference| (cycles) -> Pearson correlation and the least-squares slope through the origin."""
    n = min(len(out), len(ref)) // (sr_frames * win)
    e, p = [], []
    for i in range(n):
        s = slice(i * sr_frames * win, (i + 1) * sr_frames * win)
        den = np.sqrt((ref[s].astype(np.float64) ** 2).mean())
        if den < 50:       # silence
            continue
        e.append(np.sqrt(((out[s].astype(np.float64) - ref[s]) ** 2).mean()) / den)
        # the output is cut t_pad_tgt samples in: frame index of the window in the padded track
        f0 = 300 + i * win
        p.append(np.abs(drift[f0:f0 + win]).mean())
    e, p = np.array(e), np.array(p)
    return {"windows": int(len(e)), "pearson_r": float(np.corrcoef(e, p)[0, 1]),
            "slope_rel_error_per_cycle": float((e * p).sum() / (p * p).sum()),
            "median_rel_error": float(np.median(e)), "median_abs_phase_cycles": float(np.median(p))}


def monte_carlo(sig_hz, slope, frames, trials=20000, seed=0):
    """Waveform relative RMS between two evaluations whose f0 differs by unbiased white noise of `sig_hz` per frame: the phase
    difference is a random walk (10 ms per frame), the waveform error follows it with `slope` (relative error per cycle)."""
    rng = np.random.default_rng(seed)
    q = []
    for i in range(0, trials, 2000):
        w = np.cumsum(rng.normal(0.0, sig_hz * 0.01, size=(2000, frames)), axis=1)
        q.append(slope * np.sqrt((w ** 2).mean(axis=1)))
    q = np.concatenate(q)
    return {"median": float(np.median(q)), "p90": float(np.quantile(q, 0.9)), "p95": float(np.quantile(q, 0.95)),
            "p99": float(np.quantile(q, 0.99)), "p999": float(np.quantile(q, 0.999)), "mean": float(q.mean())}, q


def main():
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    tracks_path = sys.argv[sys.argv.index("--tracks") + 1] if "--tracks" in sys.argv else None
    import torch
    import conftest
    conftest._bind("hip")
    from test_pipeline import build, noise_fn_for
    from synthetic import weights
    from synthetic.inputs import vocal_like
    gold = np.load(os.path.join(ROOT, "tests/golden/pipeline_c1_30s.npz"))
    g64 = np.load(os.path.join(ROOT, "tests/golden/pipeline_c1_30s_fp64.npz"))
    seed, x = int(gold["seed"][0]), tuple(int(v) for v in gold["x"])
    nets = weights.full_model_set(seed)
    audio = vocal_like(float(gold["seconds"][0]), 16000, seed + 5)
    vc, hub, net_g, tgt_sr = build(conftest.Dev("hip"), nets, x)
    f0_hip = np.zeros(len(gold["f0"]))
    seen = []

    def capture(lo, hi, f0):
        arr = f0.detach().cpu().numpy() if torch.is_tensor(f0) else np.asarray(f0)
        m = min(hi, len(f0_hip)) - lo
        f0_hip[lo:lo + m] = arr[:m]
        seen.append((lo, hi))
        return f0
    vc._estimated_f0 = capture
    out = vc.pipeline(hub, net_g, 0, audio, "x.wav", [0, 0, 0], 0, "rmvpe", "", 0.5, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128,
                      noise_fn=noise_fn_for(nets))
    res = {"schedule": "progressive" if vc.last_profile["f0_progressive"] else "one launch", "f0_ranges": len(seen),
           "frames": int(len(f0_hip))}
    drifts = {}
    res["pairs"] = []
    for name, a, b in (("hip - fp64", f0_hip, g64["f0"]), ("reference - fp64", gold["f0"], g64["f0"]), ("hip - reference", f0_hip, gold["f0"])):
        st, drifts[name] = pair_stats(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), name)
        res["pairs"].append(st)
    ref = gold["audio"]
    d = np.abs(out.astype(np.int64) - ref.astype(np.int64))
    res["waveform_hip_vs_reference"] = {"rel_rms": float(np.sqrt((d.astype(np.float64) ** 2).sum() / (ref.astype(np.float64) ** 2).sum())),
                                        "max_lsb": int(d.max()), "le1": float((d <= 1).mean())}
    fol = follow(out, ref, drifts["hip - reference"])
    res["waveform_follows_phase_hip_vs_reference"] = fol
    # ---- the noise model and what it says about the distances observed
    slope = fol["slope_rel_error_per_cycle"]
    n = len(f0_hip)
    sig = {p["pair"]: float(np.std(np.asarray(a, dtype=np.float64)[:n] - np.asarray(b, dtype=np.float64)[:n], ddof=1))
           for p, (a, b) in zip(res["pairs"], ((f0_hip, g64["f0"]), (gold["f0"], g64["f0"]), (f0_hip, gold["f0"])))}
    model = {}
    for name, observed in (("hip - reference", res["waveform_hip_vs_reference"]["rel_rms"]),
                           ("reference - fp64", float(g64["ref_rel_rms"][0]))):
        qs, q = monte_carlo(sig[name], slope, n)
        model[name] = {"f0_sigma_hz_per_frame": sig[name], "waveform_rel_rms_quantiles": qs, "observed": observed,
                       "observed_percentile": float((q < observed).mean())}
    res["noise_model"] = model
    if "--samples" in sys.argv:
        from aicovergen_amd import ops

        def distance(o, r, decim=1):
            d = np.abs(o[::decim].astype(np.int64) - r.astype(np.int64))
            return {"rel_rms": float(np.sqrt((d.astype(np.float64) ** 2).sum() / (r.astype(np.float64) ** 2).sum())),
                    "max_lsb": int(d.max()), "le1": float((d <= 1).mean())}

        def run(aud, **env):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                cap = np.zeros(n)

                def capture2(lo, hi, f0):
                    arr = f0.detach().cpu().numpy() if torch.is_tensor(f0) else np.asarray(f0)
                    m = min(hi, n) - lo
                    cap[lo:lo + m] = arr[:m]
                    return f0
                vc._estimated_f0 = capture2
                o = vc.pipeline(hub, net_g, 0, aud, "x.wav", [0, 0, 0], 0, "rmvpe", "", 0.5, 1, 3, tgt_sr, 0, 0.25, "v2", 0.33, 128,
                                noise_fn=noise_fn_for(nets))
                return o, cap
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        samples = []
        # (a) the C1 input under other equally valid summation orders of the f0 path
        o1, f1 = run(audio, AICG_F0_SEGMENTS="1")
        samples.append(dict(sample="C1 input, one-launch schedule (classifier GEMM over the whole track)", **distance(o1, ref),
                            phase_end_vs_reference_cycles=float(np.sum(f1 - gold["f0"][:n]) * 0.01)))
        old2 = ops.GRU_TWO_WORKGROUPS
        ops.GRU_TWO_WORKGROUPS = False
        try:
            o2, f2 = run(audio)
        finally:
            ops.GRU_TWO_WORKGROUPS = old2
        samples.append(dict(sample="C1 input, single-workgroup GRU kernel (another summation order of the recurrence)", **distance(o2, ref),
                            phase_end_vs_reference_cycles=float(np.sum(f2 - gold["f0"][:n]) * 0.01)))
        # (b) other inputs through the same networks, against the reference's own run on them
        for a_seed in range(2001, 2033):
            path = os.path.join(ROOT, "tests/golden/pipeline_c1_30s_audio%d.npz" % a_seed)
            if not os.path.exists(path):
                continue
            g = np.load(path)
            aud = vocal_like(float(g["seconds"][0]), 16000, a_seed + 5)
            o, f = run(aud)
            m = min(n, len(g["f0"]))
            dhz = f[:m] - g["f0"][:m]
            # frames where the two f0 values are not the same estimate at all: the salience argmax (or the 0.03 voicing threshold) fell
            # the other way on a near-tie -- listed with this implementation's top-1 - top-2 salience margin
            flips = np.nonzero(np.abs(f[:m] / np.maximum(g["f0"][:m], 1e-9) - 1) > 1e-3)[0]
            rec = dict(sample="input seed %d vs the reference's run on it" % a_seed, **distance(o, g["audio"], int(g["decim"][0])),
                       argmax_flip_frames=int(len(flips)))
            if len(flips):
                _, ap, _, _ = vc.plan(aud)
                rm = vc.model_rmvpe
                sal = rm.mel2hidden(rm.mel_extractor(ap.float()[None].to(rm.device), center=True))[0].cpu().numpy()
                top2 = np.sort(sal[flips], axis=1)[:, -2:]
                rec["flips"] = [dict(frame=int(t), f0_hip=float(f[t]), f0_reference=float(g["f0"][t]), salience_top1=float(a[1]),
                                     top1_minus_top2=float(a[1] - a[0])) for t, a in zip(flips[:12], top2[:12])]
            else:
                rec.update(f0_rel_rms=float(np.sqrt(np.mean((f[:m] / g["f0"][:m] - 1) ** 2))), f0_sigma_hz=float(dhz.std(ddof=1)),
                           f0_mean_hz=float(dhz.mean()), phase_end_vs_reference_cycles=float(dhz.sum() * 0.01))
            samples.append(rec)
        res["independent_samples"] = samples
        ends = np.array([s["phase_end_vs_reference_cycles"] for s in samples if "f0_sigma_hz" in s] + [res["pairs"][2]["phase_end_of_chunk_cycles"]])
        sg = np.array([s["f0_sigma_hz"] for s in samples if "f0_sigma_hz" in s] + [sig["hip - reference"]])
        walk = float(np.sqrt(np.mean(sg ** 2)) * 0.01 * np.sqrt(n))
        res["across_inputs"] = {
            "inputs_without_argmax_flips": int(len(ends)), "phase_end_cycles": [float(e) for e in ends],
            "mean_phase_end_cycles": float(ends.mean()), "unbiased_walk_rms_endpoint_cycles": walk,
            "mean_over_standard_error": float(ends.mean() / (walk / np.sqrt(len(ends)))),
            "waveform_rel_rms": sorted(float(s["rel_rms"]) for s in samples if "f0_sigma_hz" in s) + []}
    line = json.dumps(res)
    print(line)
    if out_path:
        with open(out_path, "w") as f:
            f.write(json.dumps(res, indent=1) + "\n")
    if tracks_path:
        np.savez_compressed(tracks_path, f0_hip=f0_hip, f0_reference=gold["f0"], f0_fp64=g64["f0"], audio_hip_decim8=out[::8])


if __name__ == "__main__":
    main()
