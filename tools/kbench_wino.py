"""MDX-Net TFC 3 x 3 layers (batch 16): Winograd F(2, 3)-along-rows kernel (conv_ws3w.h) against the direct implicit GEMM."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
if os.environ.get("AICG_LIB"):
    _lib._use_library_for_tests(os.environ["AICG_LIB"], "hip")
dev = torch.device("cuda:0")


def timeit(fn, iters=4, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for lvl, (c, t, f) in enumerate([(48, 256, 3072), (96, 128, 1536), (144, 64, 768), (192, 32, 384), (240, 16, 192)]):
    x = torch.randn(16, c, t, f, device=dev)
    w = torch.randn(c, c, 3, 3, device=dev) * 0.05
    b = torch.randn(c, device=dev) * 0.1
    pc = ops.PackedConv(w, b, padding=1, device=dev)
    out = torch.empty_like(x)
    fl = 2.0 * 16 * c * c * 9 * t * f
    ops.winograd_min_positions = 1 << 60
    ref = ops.conv(x, pc, act=ops.ACT_RELU)
    td = timeit(lambda: ops.conv(x, pc, act=ops.ACT_RELU, out=out))
    ops.winograd_min_positions = 1
    tw = timeit(lambda: ops.conv(x, pc, act=ops.ACT_RELU, out=out))
    err = ((out - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()
    print(f"L{lvl} c{c} {t}x{f}: direct {td*1e3:7.3f} ms {fl/td/1e12:6.1f} TF | winograd {tw*1e3:7.3f} ms {fl/tw/1e12:6.1f} TF-equivalent "
          f"({fl/1.5/tw/1e12:5.1f} executed; rel diff {err:.0e}) | x{td/tw:.2f}", flush=True)
