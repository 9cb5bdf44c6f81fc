"""2-D conv micro-benchmark on the GPU box (MDX-Net TFC / RMVPE levels); A/B through the AICG_CONV_* env switches."""
import os
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def case(name, n, ci, co, H, W, k=3):
    x = torch.randn(n, ci, H, W, device=dev)
    pc = ops.PackedConv(torch.randn(co, ci, k, k) * 0.05, torch.randn(co), padding=k // 2, device=dev)
    out = torch.empty(n, co, H, W, device=dev)
    t = timeit(lambda: ops.conv(x, pc, out=out, act=ops.ACT_RELU))
    print(f"{name:18s} {t*1e3:8.3f} ms {2.0*n*co*ci*k*k*H*W/t/1e12:7.1f} TF", flush=True)


print({k: v for k, v in os.environ.items() if k.startswith("AICG_")})
case("mdx_L0_c48", 16, 48, 48, 256, 3072)
case("mdx_L0_in4", 16, 4, 48, 256, 3072, k=1)
case("mdx_L1_c96", 16, 96, 96, 128, 1536)
case("mdx_L2_c144", 16, 144, 144, 64, 768)
case("mdx_L3_c192", 16, 192, 192, 32, 384)
case("mdx_L4_c240", 16, 240, 240, 16, 192)
case("mdx_mid_c288", 16, 288, 288, 8, 96)
case("rmvpe_L0_c16", 1, 16, 16, 24608, 128)
case("rmvpe_L1_c32", 1, 32, 32, 12304, 64)
case("rmvpe_L2_c64", 1, 64, 64, 6152, 32)
case("rmvpe_L3_c128", 1, 128, 128, 3076, 16)
case("rmvpe_L4_c256", 1, 256, 256, 1538, 8)
