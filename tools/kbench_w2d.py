"""MDX-Net TFC 3 x 3 layers (batch 16): direct implicit GEMM vs Winograd F(2, 3) along rows (conv_ws3w.h) vs the two-dimensional
F(2 x 2, 3 x 3) kernel (conv_w2d.h) in its eight- and four-wave forms; error of each against the direct kernel and torch (corner map)."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
if os.environ.get("AICG_LIB"):
    _lib._use_library_for_tests(os.environ["AICG_LIB"], "hip")
dev = torch.device("cuda:0")
N = int(os.environ.get("KB_N", "16"))


def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for lvl, (c, t, f) in enumerate([(48, 256, 3072), (96, 128, 1536), (144, 64, 768), (192, 32, 384), (240, 16, 192)]):
    x = torch.randn(N, c, t, f, device=dev)
    w = torch.randn(c, c, 3, 3, device=dev) * 0.05
    b = torch.randn(c, device=dev) * 0.1
    pc = ops.PackedConv(w, b, padding=1, device=dev)
    out = torch.empty_like(x)
    fl = 2.0 * N * c * c * 9 * t * f
    xc = x[:2, :, :10, :72].cpu()
    tl = F.relu(F.conv2d(F.pad(xc, (1, 0, 1, 0)), w.cpu(), b.cpu()))[:, :, :9, :71]   # top-left corner map (zero pad above / left only)
    ops.winograd_min_positions = 1 << 60
    ref = ops.conv(x, pc, act=ops.ACT_RELU)
    td = timeit(lambda: ops.conv(x, pc, act=ops.ACT_RELU, out=out))
    row = [f"L{lvl} c{c} {t}x{f} N{N}: direct {td*1e3:7.3f} ms {fl/td/1e12:6.1f} TF"]
    ops.winograd_min_positions = 1
    forms = [("1d", False, 8, False, 1.5, 0), ("2d/8w", True, 8, False, 2.25, 0), ("2d/4w", True, 4, False, 2.25, 0),
             ("2d/8w/q", True, 8, True, 2.25, 0), ("2d/4w/q", True, 4, True, 2.25, 0)]
    for name, two_d, waves, quads, macs, code in forms:
        ops.winograd2d, ops.winograd2d_waves, ops.winograd2d_quads, ops.winograd2d_code = two_d, waves, quads, code
        out.zero_()
        tw = timeit(lambda: ops.conv(x, pc, act=ops.ACT_RELU, out=out))
        err = ((out - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()
        o = out[:2, :, :9, :71].cpu()
        et = ((o - tl).pow(2).sum() / tl.pow(2).sum()).sqrt().item()
        row.append(f"{name} {tw*1e3:7.3f} ms {fl/tw/1e12:6.1f} TF-eq ({fl/macs/tw/1e12:5.1f} exec; vs direct {err:.1e}, vs torch {et:.1e}) x{td/tw:.2f}")
    print(" | ".join(row), flush=True)
