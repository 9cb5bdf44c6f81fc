"""The LDS-DMA staged k-tap 1-D convolution (csrc/conv_g1k.h) against the producer / consumer kernels on the vocoder's ResBlock layers
(one 66 s chunk at 40 kHz: 256 ch x 73 080, 128 x 730 800, 64 x 1 461 600; x + conv(lrelu(x))), ROUND-ROBIN per shape;
aicg_conv_desc.gemm_tile: 1 = conv_ws3, 12 / 13 = the 128 x 256 / 64 x 256 tile of conv_g1k (development library only:
AICG_LIB=dev), 0 = the library's policy."""
import os, sys, statistics, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
NAMES = {1: "ws3", 0: "policy", 12: "128x256", 13: "64x256"}
codes = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,12,13,0").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for c, t in [(256, 73080), (128, 730800), (64, 1461600)]:
    for k, d in [(3, 1), (3, 5), (7, 1), (7, 5), (11, 1), (11, 5)]:
        x = torch.randn(1, c, t, device=dev)
        pc = ops.PackedConv(torch.randn(c, c, k) * 0.03, torch.randn(c), padding=(k - 1) * d // 2, dilation=d, device=dev)
        out = torch.empty_like(x)
        fn = lambda: ops.conv(x, pc, out=out, res=x, pre_act=ops.ACT_LRELU, pre_slope=0.1)
        ran = {}
        for code in codes:
            ops.gemm_tile = code
            for _ in range(2): fn()
            ran[code] = _lib.last_launch()[5:8]
        torch.cuda.synchronize()
        times = {code: [] for code in codes}
        for r in range(rounds):
            for code in (codes if r % 2 == 0 else codes[::-1]):
                ops.gemm_tile = code
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3): fn()
                e1.record(); torch.cuda.synchronize()
                times[code].append(e0.elapsed_time(e1) / 3)
        ops.gemm_tile = 0
        fl = 2.0 * c * c * k * t
        print(f"C{c}@{t} k{k} d{d}: " + " | ".join(f"{NAMES[code]:>7s}[{ran[code]}] {statistics.median(v)*1e3:7.1f} us {fl/statistics.median(v)/1e9:5.1f}" for code, v in times.items()), flush=True)
        del x, out
