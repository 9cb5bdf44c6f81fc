"""MDX-Net's up-sampling layers (ConvTranspose2d k = s = 2 as a 1 x 1 GEMM + pixel-shuffle scatter + bias + ReLU + multiplicative skip, one
kernel: conv_g1 SHUF) on the three forced tiles, round-robin: short K (96 .. 192), 2.4 GB written and 2.4 GB of skip read per launch."""
import os, sys, statistics, torch
os.environ.setdefault("AICG_DEV", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
for ci, co, h, w in ((96, 48, 128, 1536), (144, 96, 64, 768), (192, 144, 32, 384)):
    x = torch.randn(16, ci, h, w, device=dev)
    pt = ops.PackedConvTranspose(torch.randn(ci, co, 2, 2, device=dev) * 0.05, torch.randn(co, device=dev) * 0.1, stride=2, device=dev)
    skip = torch.randn(16, co, 2 * h, 2 * w, device=dev)
    out = torch.empty_like(skip)
    times = {t: [] for t in (0, 2, 3, 4)}
    for r in range(5):
        for t in (0, 2, 3, 4) if r % 2 == 0 else (4, 3, 2, 0):
            ops.gemm_tile = t
            for _ in range(2): ops.conv_transpose(x, pt, act=ops.ACT_RELU, mul=skip, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.conv_transpose(x, pt, act=ops.ACT_RELU, mul=skip, out=out)
            e1.record(); torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / 3)
    ops.gemm_tile = 0
    gb = (x.numel() + 2 * skip.numel()) * 4 / 1e9
    print(f"C{ci}>{co} {h}x{w}: " + " | ".join(f"tile {t}: {statistics.median(v) * 1e3:7.1f} us ({gb / statistics.median(v):5.2f} TB/s)" for t, v in times.items()), flush=True)
