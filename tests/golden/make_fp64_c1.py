"""TEST INFRASTRUCTURE: BASELINE C1 (30 s, full-size networks, x = 3, 10, 60, 65) through the ORACLE pipeline evaluated in float64
-- every parameter, activation and table in double precision, the same seeded weights, input and noise draws as
tests/golden/pipeline_c1_30s.npz (the reference's own fp32 run).  The result stands in for exact arithmetic: the distance of the
REFERENCE's fp32 output from it is the noise floor any fp32 implementation of this pipeline lives on, and the tolerance the HIP
path is gated with (tests/test_bench_sizes.py::test_c1_pipeline_vs_reference_golden).

    python tests/golden/make_fp64_c1.py            # ~10 min of host CPU; writes tests/golden/pipeline_c1_30s_fp64.npz

How the oracle is switched to float64 without a second copy of it: torch's default dtype is set to float64, every state-dict tensor
is converted, and `Tensor.float` is rebound to `Tensor.double` for the duration of the run (the oracle spells its casts `.float()`).
Stored (decimated by 4 to keep the fixture small; statistics over 300 k samples): the float64 run's int16 output, its f0 /
coarse bins in full, and the reference-vs-float64 distances measured here against the committed golden."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
DECIM = 4


def to_double(obj):
    if torch.is_tensor(obj):
        return obj.double() if obj.is_floating_point() else obj
    if isinstance(obj, dict):
        return {k: to_double(v) for k, v in obj.items()}
    return obj


def run_fp64(nets, geo_x, audio, tgt_sr):
    from oracle import pipeline as opipe
    old_default, old_float = torch.get_default_dtype(), torch.Tensor.float
    torch.set_default_dtype(torch.float64)
    torch.Tensor.float = torch.Tensor.double
    try:
        nets64 = {k: to_double(v) for k, v in nets.items()}
        geo = opipe.Geometry(tgt_sr, *geo_x)
        return opipe.vc_pipeline(nets64, geo, audio.astype(np.float64), tgt_sr=tgt_sr)
    finally:
        torch.set_default_dtype(old_default)
        torch.Tensor.float = old_float


def distances(a, b):
    """int16 waveforms -> (relative RMS, max |diff| in LSB, share within 1 LSB, share exact)."""
    d = np.abs(a.astype(np.int64) - b.astype(np.int64))
    return (float(np.sqrt(np.sum(d.astype(np.float64) ** 2) / np.sum(b.astype(np.float64) ** 2))), int(d.max()),
            float((d <= 1).mean()), float((d == 0).mean()))


def main():
    from synthetic import weights
    from synthetic.inputs import vocal_like
    small = "--small" in sys.argv          # plumbing check on the miniature networks (seconds)
    if small:
        gold = np.load(os.path.join(HERE, "pipeline_small_2p6s.npz"))
        seed, x, seconds = int(gold["seed"][0]), (1, 1, 1, 2), float(gold["seconds"][0])
        nets = weights.small_model_set(seed)
    else:
        gold = np.load(os.path.join(HERE, "pipeline_c1_30s.npz"))
        seed, x, seconds = int(gold["seed"][0]), tuple(int(v) for v in gold["x"]), float(gold["seconds"][0])
        nets = weights.full_model_set(seed)
    torch.set_num_threads(os.cpu_count() or 1)
    audio = vocal_like(seconds, 16000, seed + 5)
    t0 = time.time()
    out, info = run_fp64(nets, x, audio, nets["synth_cfg"][-1])
    took = time.time() - t0
    ref = gold["audio"]
    assert out.shape == ref.shape, (out.shape, ref.shape)
    rel, mx, le1, ex = distances(ref, out)
    print("reference (fp32) vs float64: rel rms %.3e, max %d LSB of peak %d, <= 1 LSB on %.4f, exact on %.4f  [%.0f s]"
          % (rel, mx, int(np.abs(out).max()), le1, ex, took))
    extra = {}
    if "f0" in gold.files:
        n = min(len(gold["f0"]), len(info["f0"]))
        v = (gold["f0"][:n] > 0) & (info["f0"][:n] > 0)
        f0rel = np.abs(gold["f0"][:n][v] / info["f0"][:n][v] - 1)
        bins = int(np.sum(gold["coarse"][:n] != info["coarse"][:n]))
        print("reference f0 vs float64: relative rms %.3e, max %.3e over %d voiced frames; %d coarse bins differ"
              % (np.sqrt(np.mean(f0rel ** 2)), f0rel.max(), int(v.sum()), bins))
        extra = dict(ref_f0_rel_rms=np.array([np.sqrt(np.mean(f0rel ** 2))]), ref_f0_rel_max=np.array([f0rel.max()]),
                     ref_coarse_bins_differ=np.array([bins]))
    if small:
        return
    np.savez_compressed(os.path.join(HERE, "pipeline_c1_30s_fp64.npz"), decim=np.array([DECIM]),
                        audio=out[::DECIM],
                        f0=np.asarray(info["f0"], np.float64), coarse=np.asarray(info["coarse"]).astype(np.int16),
                        ref_rel_rms=np.array([rel]), ref_max_lsb=np.array([mx]), ref_le1=np.array([le1]), ref_exact=np.array([ex]),
                        fp64_cpu_seconds=np.array([took]), **extra)


if __name__ == "__main__":
    main()
