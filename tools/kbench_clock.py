"""Is the conv kernel clock-/power-limited?  Same launch on random and on zero-filled operands (DVFS give-back,
MI355X_MICROARCH.md): a large gap means the MFMA loop already runs at the clock the power budget allows."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")


def timeit(fn, iters=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def case(name, n, ci, co, H, W, k=3, mode="randn"):
    x = torch.randn(n, ci, H, W, device=dev)
    w = torch.randn(co, ci, k, k) * 0.05
    if mode == "zeros":
        x.zero_(); w.zero_()
    elif mode == "small":   # few mantissa bits toggling
        x = torch.round(x * 2) / 2; w = torch.round(w * 64) / 64
    pc = ops.PackedConv(w, torch.zeros(co), padding=k // 2, device=dev)
    out = torch.empty(n, co, H, W, device=dev)
    t = timeit(lambda: ops.conv(x, pc, out=out, act=ops.ACT_RELU))
    print(f"{name:14s} {mode:6s} {t*1e3:8.3f} ms {2.0*n*co*ci*k*k*H*W/t/1e12:7.1f} TF", flush=True)


for mode in ("randn", "zeros", "small", "randn"):
    case("mdx_L1_c96", 16, 96, 96, 128, 1536, mode=mode)
    case("mdx_L2_c144", 16, 144, 144, 64, 768, mode=mode)
