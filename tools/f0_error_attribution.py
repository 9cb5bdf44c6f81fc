"""Which stage of RMVPE produces the fp32 f0 error?  (round 5, behind tools/c1_f0_bias.py: over 26 inputs the HIP and the reference f0 tracks
differ by a systematic 7e-9 relative -- 1/8 of an fp32 ulp -- that the vocoder's source integrates into a waveform distance.)

CPU only (torch): the BASELINE C1 input through the oracle's RMVPE with the seeded full-size weights, in float64 except ONE stage in
float32 -- log-mel front end / U-Net + output conv / BiGRU / classifier + sigmoid -- and everything in float32.  Per variant: relative f0
error against the all-float64 track (rms, signed mean, signed mean per f0 tercile) and the end-of-chunk phase it integrates to.

    python tools/f0_error_attribution.py > profiles/r05_f0_error_attribution.json
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rmvpe as orm          # noqa: E402
from synthetic import weights            # noqa: E402
from synthetic.inputs import vocal_like  # noqa: E402


def stage_mel(audio, dt):
    basis = torch.from_numpy(orm.mel_filterbank()).to(dt)
    a = torch.from_numpy(audio).to(dt).unsqueeze(0)
    fft = torch.stft(a, n_fft=1024, hop_length=160, win_length=1024, window=torch.hann_window(1024, dtype=dt), center=True, return_complex=True)
    mag = torch.sqrt(fft.real.pow(2) + fft.imag.pow(2))
    mel = torch.log(torch.clamp(torch.matmul(basis, mag), min=1e-5))
    n = mel.shape[-1]
    return F.pad(mel, (0, 32 * ((n - 1) // 32 + 1) - n), mode="reflect"), n


def stage_unet(sd, mel):
    """oracle e2e_forward up to the GRU input (1, T, 384), in the dtype of `sd`."""
    x = mel.transpose(-1, -2).unsqueeze(1)
    x = orm._bn_eval(x, sd, "unet.encoder.bn")
    skips = []
    for i in range(orm._count(sd, "unet.encoder.layers.%d.")):
        for b in range(orm._count(sd, "unet.encoder.layers.%d.conv." % i + "%d.")):
            x = orm.conv_block_res(sd, "unet.encoder.layers.%d.conv.%d" % (i, b), x)
        skips.append(x)
        x = F.avg_pool2d(x, 2)
    for i in range(orm._count(sd, "unet.intermediate.layers.%d.")):
        for b in range(orm._count(sd, "unet.intermediate.layers.%d.conv." % i + "%d.")):
            x = orm.conv_block_res(sd, "unet.intermediate.layers.%d.conv.%d" % (i, b), x)
    for i in range(orm._count(sd, "unet.decoder.layers.%d.")):
        p = "unet.decoder.layers.%d." % i
        x = F.conv_transpose2d(x, sd[p + "conv1.0.weight"], None, stride=2, padding=1, output_padding=1)
        x = F.relu(orm._bn_eval(x, sd, p + "conv1.1"))
        x = torch.cat((x, skips[-1 - i]), dim=1)
        for b in range(orm._count(sd, p + "conv2.%d.")):
            x = orm.conv_block_res(sd, p + "conv2.%d" % b, x)
    x = F.conv2d(x, sd["cnn.weight"], sd["cnn.bias"], padding=1)
    return x.transpose(1, 2).flatten(-2)


def stage_gru(sd, x):
    dt = x.dtype
    g = torch.nn.GRU(384, 256, num_layers=1, batch_first=True, bidirectional=True).to(dt)
    g.load_state_dict({k[len("fc.0.gru."):]: v for k, v in sd.items() if k.startswith("fc.0.gru.")})
    return g(x)[0]


def stage_fc(sd, x):
    return torch.sigmoid(F.linear(x, sd["fc.1.weight"], sd["fc.1.bias"]))


def run(sd64, sd32, audio, fp32_stages):
    def dt(stage):
        return torch.float32 if stage in fp32_stages else torch.float64

    def sd(stage):
        return sd32 if stage in fp32_stages else sd64
    with torch.no_grad():
        mel, n = stage_mel(audio.astype(np.float64 if dt("mel") == torch.float64 else np.float32), dt("mel"))
        x = stage_unet(sd("unet"), mel.to(dt("unet")))
        x = stage_gru(sd("gru"), x.to(dt("gru")))
        s = stage_fc(sd("fc"), x.to(dt("fc")))
    sal = s[0, :n].to(torch.float32 if "fc" in fp32_stages or fp32_stages == {"all"} else torch.float64).numpy()
    return orm.decode(sal.astype(np.float64) if sal.dtype == np.float64 else sal, 0.03)


def main():
    torch.set_num_threads(8)
    nets = weights.full_model_set(1234)
    sd32 = {k: v.float() for k, v in nets["rmvpe_sd"].items() if torch.is_tensor(v)}
    sd64 = {k: v.double() for k, v in sd32.items()}
    audio = vocal_like(30.0, 16000, 1234 + 5)
    pad = np.pad(audio, (48000, 48000), mode="reflect")          # the pipeline's x_pad = 3 s reflect padding (C1: one chunk)
    ref = run(sd64, sd32, pad, set())
    q = np.quantile(ref[ref > 0], [0, 1 / 3, 2 / 3, 1])
    out = {"frames": int(len(ref)), "f0_terciles_hz": [float(v) for v in q], "variants": {}}
    for name, st in (("mel in fp32", {"mel"}), ("U-Net in fp32", {"unet"}), ("BiGRU in fp32", {"gru"}), ("classifier + sigmoid in fp32", {"fc"}),
                     ("everything in fp32 (torch CPU)", {"mel", "unet", "gru", "fc"})):
        f0 = run(sd64, sd32, pad, st)
        v = (f0 > 0) & (ref > 0)
        r = f0[v] / ref[v] - 1
        rec = {"rel_rms": float(np.sqrt((r ** 2).mean())), "rel_mean_signed": float(r.mean()),
               "rel_mean_standard_error": float(r.std(ddof=1) / np.sqrt(len(r))),
               "phase_end_cycles": float(np.sum(np.where(v, f0 - ref, 0.0)) * 0.01), "voicing_or_argmax_flips": int(np.sum(np.abs(r) > 1e-3))}
        for i in range(3):
            m = v & (ref >= q[i]) & (ref <= q[i + 1])
            rr = f0[m] / ref[m] - 1
            rec["tercile_%d_mean_signed" % i] = float(rr.mean())
            rec["tercile_%d_standard_error" % i] = float(rr.std(ddof=1) / np.sqrt(len(rr)))
        out["variants"][name] = rec
        print(name, json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
