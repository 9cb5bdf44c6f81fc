"""The one-dimensional Winograd F(2, 3) kernel (csrc/conv_g1w.h) against the producer / consumer kernels on the vocoder's dilation-1 ResBlock
layers (one 66 s chunk at 40 kHz: 256 ch x 73 080, 128 x 730 800, 64 x 1 461 600, 32 x 2 923 200; x + conv(lrelu(x))), ROUND-ROBIN per shape.
Columns: ws3 = the direct kernels (ops.winograd1d off), then conv_g1w's tiles (aicg_conv_desc.gemm_tile 2 / 3 / 4 = 64 x 256 / 32 x 512 /
128 x 128) and its policy (0).  TF = direct-form TFLOP/s (the Winograd kernel executes 4/6, 10/14, 15/22 of them)."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
NAMES = {-1: "ws3", 0: "policy", 2: "64x256", 3: "32x512", 5: "32x512s", 6: "pers"}
codes = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "-1,3,5").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dils = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1").split(",")]
tot = {c: 0.0 for c in codes}
for c, t in [(256, 73080), (128, 730800), (64, 1461600), (32, 2923200)]:
  for k in (3, 7, 11):
    for dil in dils:
        x = torch.randn(1, c, t, device=dev)
        pc = ops.PackedConv(torch.randn(c, c, k) * 0.03, torch.randn(c), padding=(k - 1) // 2 * dil, dilation=dil, device=dev)
        out = torch.empty_like(x)

        def fn(code):
            ops.winograd1d = code >= 0
            ops.gemm_tile = max(code, 0)
            ops.conv(x, pc, out=out, res=x, pre_act=ops.ACT_LRELU, pre_slope=0.1)
        ran, ref = {}, None
        for code in codes:
            for _ in range(2): fn(code)
            ran[code] = _lib.last_launch()[5:9]
            torch.cuda.synchronize()
            if ref is None: ref = out.clone()
            else: ran[code] += " %.0e" % ((out - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()
        times = {code: [] for code in codes}
        for r in range(rounds):
            for code in (codes if r % 2 == 0 else codes[::-1]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3): fn(code)
                e1.record(); torch.cuda.synchronize()
                times[code].append(e0.elapsed_time(e1) / 3)
        fl = 2.0 * c * c * k * t
        row = [f"C{c:3d} k{k:2d} d{dil}"]
        for code in codes:
            ms = statistics.median(times[code])
            tot[code] += ms
            row.append(f"{NAMES[code]:7s} {ms*1e3:7.1f} us {fl/ms/1e9:6.1f} TF [{ran[code]}]")
        print(" | ".join(row), flush=True)
ops.gemm_tile, ops.winograd1d = 0, True
print("sum over the layers: " + " | ".join(f"{NAMES[c]} {tot[c]:.3f} ms" for c in codes))
