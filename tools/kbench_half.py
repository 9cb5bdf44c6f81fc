"""fp32 against fp16-operand runs (aicg_conv_desc.split == 2, ops.mark_half) of the layers that have the form: the vocoder's ResBlock
layers of a 66 s chunk (conv_g1w) and HuBERT's / the vocoder's 1 x 1 GEMMs (conv_g1), round-robin."""
import os, sys, statistics, torch
os.environ.setdefault("AICG_DEV", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
rounds = int(os.environ.get("KB_ROUNDS", "5"))


def bench(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3


tot = {False: 0.0, True: 0.0}
layers = [("vocoder C%d k%d d%d" % (c, k, d), c, c, k, d, t, True) for c, t in ((256, 66000), (128, 660000), (64, 1320000), (32, 2640000)) for k in (3, 7, 11) for d in (1, 3)]
layers += [("hubert 768->3072", 768, 3072, 1, 1, 3300, False), ("hubert 3072->768", 3072, 768, 1, 1, 3300, False), ("hubert 768->768", 768, 768, 1, 1, 3300, False),
           ("vocoder up 512->2560", 512, 2560, 1, 1, 6600, False)]
for name, ci, co, k, d, t, res in layers:
    x = torch.randn(1, ci, t, device=dev)
    pc = ops.PackedConv(torch.randn(co, ci, k, device=dev) * 0.05, torch.randn(co, device=dev) * 0.1, padding=(k - 1) // 2 * d, dilation=d, device=dev)
    out = torch.empty(1, co, t, device=dev)
    times = {False: [], True: []}
    outs = {}
    for r in range(rounds):
        for half in ((False, True) if r % 2 == 0 else (True, False)):
            ops.mark_half(pc, half)
            f = (lambda: ops.conv(x, pc, res=x, pre_act=ops.ACT_LRELU, pre_slope=0.1, out=out)) if res else (lambda: ops.conv(x, pc, out=out))
            times[half].append(bench(f))
            outs[half] = out.clone()
    err = float(((outs[True] - outs[False]).double().pow(2).sum() / outs[False].double().pow(2).sum()).sqrt())
    m32, m16 = statistics.median(times[False]), statistics.median(times[True])
    fl = 2.0 * ci * co * k * t
    tot[False] += m32; tot[True] += m16
    print(f"{name:24s} fp32 {m32 * 1e3:8.1f} us ({fl / m32 / 1e9:6.1f} TF direct) | f16 {m16 * 1e3:8.1f} us ({fl / m16 / 1e9:6.1f} TF direct) | x{m32 / m16:4.2f} | rel rms f16 vs fp32 {err:.1e}", flush=True)
print(f"sum: fp32 {tot[False]:.3f} ms | f16 {tot[True]:.3f} ms")
