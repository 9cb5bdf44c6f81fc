"""Attention in the step runs at 0.42 of the fp32 MFMA peak (stage table of the bench line) and at 0.64 stand-alone: is that the kernel or
the company it keeps?  (VERDICT r5 "next" #7.)  In the HuBERT phase the main stream's attention launches share the chip with the f0 branch
on a high-priority side stream (RMVPE's mel + U-Net, then the BiGRU segments on eight CUs).  This tool times the step's two HuBERT attention
shapes (12 heads x 64, T = 3300 per chunk: `attn_fwd<64,3>`; what the bench batches over four chunks) on the main stream

    alone | beside a looping RMVPE U-Net (high-priority side stream, as pipeline() runs it) | beside the same at default priority |
    beside the BiGRU recurrence alone (eight workgroups)

and, the other way round, what the U-Net loses beside a looping attention.  GPU box:  python tools/kbench_attn_contention.py"""
import os, sys, time, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
from aicovergen_amd.rmvpe import RMVPE  # noqa: E402
from synthetic import weights  # noqa: E402
from synthetic.inputs import vocal_like  # noqa: E402
import numpy as np  # noqa: E402

dev = torch.device("cuda:0")
H, D, T = 12, 64, 3300
q, k, v = (torch.randn(H * D, T, device=dev) * 0.3 for _ in range(3))
flops = 4.0 * T * T * H * D


def attn():
    return ops.attention(q, k, v, H)


r = RMVPE(None, False, dev, state_dict=weights.rmvpe_state_dict(weights.RMVPE_FULL, 1235))
audio = np.pad(vocal_like(60.0, 16000, seed=3), (48000, 48000), mode="reflect")
mel = r.mel_extractor(torch.from_numpy(audio).float()[None].to(dev), center=True)


def unet_gru():
    return r.mel2hidden(mel)


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    return e0, e1


for _ in range(3):
    attn(); unet_gru()
torch.cuda.synchronize()
e0, e1 = timed(attn, 40); torch.cuda.synchronize()
t_alone = e0.elapsed_time(e1) / 40
e0, e1 = timed(unet_gru, 4); torch.cuda.synchronize()
u_alone = e0.elapsed_time(e1) / 4
print("attention alone: %.3f ms per call = %.1f TFLOP/s (%.2f of 157.3)" % (t_alone, flops / t_alone / 1e9, flops / t_alone / 1e9 / 157.3))
print("RMVPE mel2hidden (U-Net + BiGRU + classifier, 66 s of audio) alone: %.2f ms per call" % u_alone, flush=True)

for prio, name in ((-1, "high priority (as pipeline() runs the f0 branch)"), (0, "default priority")):
    side = torch.cuda.Stream(device=dev, priority=prio)
    stop = [False]
    # background: the f0 branch looping on the side stream, queued far enough ahead that the stream never runs dry
    n_bg = max(2, int(60 * t_alone * 1.6 / u_alone) + 1)   # a little longer than the foreground's 60 calls
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        b0, b1 = timed(unet_gru, n_bg)
    time.sleep(0.002)
    e0, e1 = timed(attn, 60)           # main stream: 60 attention calls ~ 20 ms, well inside the background's ~12 x u_alone
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 60
    u = b0.elapsed_time(b1) / n_bg
    print("attention beside the f0 branch at %s: %.3f ms per call = %.1f TFLOP/s (%.2f; x%.2f of alone); the f0 branch meanwhile %.2f ms per call (x%.2f)"
          % (name, t, flops / t / 1e9, flops / t / 1e9 / 157.3, t / t_alone, u, u / u_alone), flush=True)

# the other direction: attention as the background, the f0 branch in front
side = torch.cuda.Stream(device=dev, priority=0)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    n_at = int(3 * u_alone * 1.6 / t_alone) + 1
    b0, b1 = timed(attn, n_at)
time.sleep(0.002)
e0, e1 = timed(unet_gru, 3)
torch.cuda.synchronize()
print("f0 branch beside looping attention (both default priority): %.2f ms per call (x%.2f of alone); attention meanwhile %.3f ms per call (x%.2f)"
      % (e0.elapsed_time(e1) / 3, e0.elapsed_time(e1) / 3 / u_alone, b0.elapsed_time(b1) / n_at, b0.elapsed_time(b1) / n_at / t_alone))
