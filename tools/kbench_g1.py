"""The LDS-DMA staged 1 x 1 GEMM (csrc/conv_g1.h) against the producer / consumer kernels it replaces, per shape and per tile, ROUND-ROBIN in
one process (a hot chip's clock sags over a sustained run: forms timed one after the other are not comparable).  Dev library;
aicg_conv_desc.gemm_tile selects the form per launch: 1 = conv_ws3 (the kernel off), 0 = the library's policy, 2 / 3 / 4 = the 128 x 256 /
64 x 256 / 192 x 256 tile, 5 / 6 = the 256 x 256 / 128 x 512 probes, 12 / 13 / 14 = 2 / 3 / 4 with a stage's DMA as one burst.
Shapes: HuBERT's per-token GEMMs at the benched token count (four chunks side by side, padded to 32) and for one rank of two, MDX-Net's
up-sampling GEMM + pixel shuffle with its multiplicative skip at every level (batch 16), enc_p / vocoder-sized GEMMs.
usage: kbench_g1.py [codes, comma separated] [rounds]"""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
NAMES = {1: "ws3", 0: "policy", 2: "128x256", 3: "64x256", 4: "192x256", 5: "256x256", 6: "128x512", 12: "128b", 13: "64b", 14: "192b"}
codes = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,0,2,3,4,12,13,14").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def ab(label, fn, flop, per=3):
    ran = {}
    for c in codes:
        ops.gemm_tile = c
        for _ in range(2): fn()
        ran[c] = _lib.last_launch()[5:8]
    torch.cuda.synchronize()
    times = {c: [] for c in codes}
    for r in range(rounds):
        for c in (codes if r % 2 == 0 else codes[::-1]):
            ops.gemm_tile = c
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per): fn()
            e1.record(); torch.cuda.synchronize()
            times[c].append(e0.elapsed_time(e1) / per)
    ops.gemm_tile = 0
    med = {c: statistics.median(v) for c, v in times.items()}
    best = min((c for c in codes if c != 0), key=lambda c: med[c])
    print(f"{label:30s} " + " | ".join(f"{NAMES[c]:>7s}[{ran[c]}] {med[c]*1e3:7.1f} us {flop/med[c]/1e9:5.1f}" for c in codes)
          + f"   best {NAMES[best]}" + (f", policy +{(med[0]/med[best]-1)*100:.0f}%" if 0 in med else ""), flush=True)


for t in (13216, 6624):
    for ci, co, act in [(768, 3072, ops.ACT_GELU), (3072, 768, ops.ACT_NONE), (768, 768, ops.ACT_NONE), (768, 1536, ops.ACT_NONE),
                        (768, 2304, ops.ACT_NONE), (512, 768, ops.ACT_NONE)]:
        x = torch.randn(1, ci, t, device=dev)
        pc = ops.PackedConv(torch.randn(co, ci, 1) * 0.03, torch.randn(co), device=dev)
        out = torch.empty(1, co, t, device=dev)
        ab(f"{ci}>{co}{'g' if act == ops.ACT_GELU else ' '}@{t}", lambda: ops.conv(x, pc, out=out, act=act), 2.0 * ci * co * t)
for ci, co, h, w in [(96, 48, 128, 1536), (144, 96, 64, 768), (192, 144, 32, 384), (240, 192, 16, 192), (288, 240, 8, 96)]:
    pt = ops.PackedConvTranspose(torch.randn(ci, co, 2, 2) * 0.05, torch.randn(co), stride=2, device=dev)
    x, skip = torch.randn(16, ci, h, w, device=dev), torch.randn(16, co, 2 * h, 2 * w, device=dev)
    out = torch.empty_like(skip)
    gb = 4.0 * (x.numel() + 2 * skip.numel()) / 1e9
    ab(f"us C{ci}>{4*co} {h}x{w} ({gb:.2f} GB)", lambda: ops.conv_transpose(x, pt, out=out, act=ops.ACT_RELU, mul=skip), 2.0 * 16 * ci * 4 * co * h * w, per=2)
    del x, skip, out
for ci, co, t in [(192, 192, 6420), (192, 384, 6420), (768, 192, 6420), (192, 768, 6420), (256, 2048, 73080), (128, 256, 756800), (64, 128, 1513600)]:
    x = torch.randn(1, ci, t, device=dev)
    pc = ops.PackedConv(torch.randn(co, ci, 1) * 0.03, torch.randn(co), device=dev)
    out = torch.empty(1, co, t, device=dev)
    ab(f"{ci}>{co}@{t}", lambda: ops.conv(x, pc, out=out), 2.0 * ci * co * t)
