"""Kernel micro-benchmarks on the GPU box (not the judged bench.py): per-shape TFLOP/s of the conv kernel
and GB/s of the STFT kernels.  Writes gpurun_out/kbench.json."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
res = {}


def timeit(fn, iters=10, warm=2):
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


def conv_case(name, ci, co, k, d, T, groups=1, stride=1, fused=True):
    x = torch.randn(1, ci, T, device=dev)
    w = torch.randn(co, ci // groups, k) * 0.05
    pc = ops.PackedConv(w, torch.randn(co), padding=(k - 1) * d // 2, dilation=d, stride=stride, groups=groups, device=dev)
    out = torch.empty(1, co, pc.out_hw(1, T)[1], device=dev)
    kw = dict(pre_act=ops.ACT_LRELU, pre_slope=0.1) if fused else {}
    res_t = x if (fused and ci == co and stride == 1) else None
    t = timeit(lambda: ops.conv(x, pc, res=res_t, out=out, **kw))
    fl = 2.0 * co * (ci // groups) * k * out.shape[-1]
    res[name] = {"ms": t * 1e3, "tflops": fl / t / 1e12}
    print(name, res[name], flush=True)


# vocoder ResBlock1 shapes for one 66 s chunk at 40 kHz
for (c, L) in [(256, 66000), (128, 660000), (64, 1320000), (32, 2640000)]:
    for k in (3, 7, 11):
        for d in (1, 5):
            conv_case(f"rb_c{c}_k{k}_d{d}", c, c, k, d, L)
# GEMM-like 1x1 (HuBERT FFN 768->3072, T=3300) and enc_p sizes
conv_case("lin_768_3072_T3300", 768, 3072, 1, 1, 3300, fused=False)
conv_case("lin_3072_768_T3300", 3072, 768, 1, 1, 3300, fused=False)
conv_case("ffn_192_768_k3_T6600", 192, 768, 3, 1, 6600, fused=False)
conv_case("hubert_fe_512_k3_s2", 512, 512, 3, 1, 105615, stride=2, fused=False)
conv_case("posconv_768_k128_g16", 768, 768, 128, 1, 3300, groups=16, fused=False)



def conv2d_case(name, n, ci, co, H, W, k=3):
    x = torch.randn(n, ci, H, W, device=dev)
    pc = ops.PackedConv(torch.randn(co, ci, k, k) * 0.05, torch.randn(co), padding=k // 2, device=dev)
    out = torch.empty(n, co, H, W, device=dev)
    t = timeit(lambda: ops.conv(x, pc, out=out, act=ops.ACT_RELU), iters=5)
    fl = 2.0 * n * co * ci * k * k * H * W
    res[name] = {"ms": t * 1e3, "tflops": fl / t / 1e12}
    print(name, res[name], flush=True)


# MDX-Net TFC convs per level (batch 16 = 8 windows x {+,-}) and RMVPE levels
conv2d_case("mdx_L0_c48", 16, 48, 48, 256, 3072)
conv2d_case("mdx_L1_c96", 16, 96, 96, 128, 1536)
conv2d_case("mdx_L2_c144", 16, 144, 144, 64, 768)
conv2d_case("mdx_L3_c192", 16, 192, 192, 32, 384)
conv2d_case("mdx_L4_c240", 16, 240, 240, 16, 192)
conv2d_case("mdx_mid_c288", 16, 288, 288, 8, 96)
conv2d_case("rmvpe_L0_c16", 1, 16, 16, 24608, 128)
conv2d_case("rmvpe_L2_c64", 1, 64, 64, 6152, 32)
conv2d_case("rmvpe_L4_c256", 1, 256, 256, 1538, 8)
conv2d_case("rmvpe_mid_c512", 1, 512, 512, 769, 4)
# TDF linear (level 0): rows = 16*48*256, K = 3072 -> 384 -> 3072
xg = torch.randn(16, 48, 256, 3072, device=dev)
w1 = torch.randn(384, 3072, device=dev) * 0.02
sc, sh = torch.ones(48, device=dev), torch.zeros(48, device=dev)
t = timeit(lambda: ops.linear_last(xg, w1, None, sc, sh, act=ops.ACT_RELU), iters=5)
res["tdf_L0_3072_384"] = {"ms": t * 1e3, "tflops": 2.0 * 16 * 48 * 256 * 3072 * 384 / t / 1e12}
print("tdf", res["tdf_L0_3072_384"], flush=True)
del xg

# STFT / iSTFT: one MDX window batch (22 windows x 2 channels), frame-major internal layout
x = torch.randn(44, 261120, device=dev)
t = timeit(lambda: ops.stft(x, 7680, 1024, 3072, frame_major=True))
by = 44 * (261120 * 4 + 2 * 3072 * 256 * 4)
res["stft_7680_b22"] = {"ms": t * 1e3, "GBps": by / t / 1e9}
sp = ops.stft(x, 7680, 1024, 3072, frame_major=True)
t = timeit(lambda: ops.istft(sp, 7680, 1024, 261120, frame_major=True))
res["istft_7680_b22"] = {"ms": t * 1e3, "GBps": by / t / 1e9}
print(res["stft_7680_b22"], res["istft_7680_b22"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "kbench.json"), "w"), indent=1)
