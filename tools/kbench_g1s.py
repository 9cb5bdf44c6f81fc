"""HuBERT's stride-2 feature-extractor layers (512 -> 512, k = 3 / 2, GELU; one 66 s chunk: 211 231 -> 105 615 -> ... -> 3 300 frames) on the
LDS-DMA staged stride-2 kernel (csrc/conv_g1s.h) against the producer / consumer kernels, ROUND-ROBIN per layer.
aicg_conv_desc.gemm_tile: 1 = conv_ws3 (the kernel these layers ran on until round 5), 2 / 3 = the 128 x 256 / 64 x 256 tile of conv_g1s,
0 = the library's policy.  Rows padded to 16 bytes on both sides, as hubert._frontend lays them out."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
NAMES = {1: "ws3", 0: "policy", 2: "128x256", 3: "64x256"}
codes = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,0").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
tot = {c: 0.0 for c in codes}
for k, t_in in [(3, 211231), (3, 105615), (3, 52807), (3, 26403), (2, 13201), (2, 6600)]:
    t_out = (t_in - k) // 2 + 1
    xb = torch.randn(1, 512, (t_in + 3) // 4 * 4, device=dev)
    x = xb[:, :, :t_in]
    pc = ops.PackedConv(torch.randn(512, 512, k) * 0.03, None, stride=2, device=dev)
    out = torch.empty(1, 512, (t_out + 3) // 4 * 4, device=dev)[:, :, :t_out]
    fn = lambda: ops.conv(x, pc, out=out, act=ops.ACT_GELU)
    ran, ref = {}, None
    for code in codes:
        ops.gemm_tile = code
        for _ in range(2): fn()
        ran[code] = _lib.last_launch()
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        else: ran[code] += " %.0e" % ((out - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()
    times = {code: [] for code in codes}
    for r in range(rounds):
        for code in (codes if r % 2 == 0 else codes[::-1]):
            ops.gemm_tile = code
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn()
            e1.record(); torch.cuda.synchronize()
            times[code].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * 512 * 512 * k * t_out
    row = [f"k{k} T_in {t_in:6d}"]
    for code in codes:
        ms = statistics.median(times[code])
        tot[code] += ms
        row.append(f"{NAMES[code]:8s} {ms*1e3:7.1f} us {fl/ms/1e9:6.1f} TF [{ran[code]}]")
    print(" | ".join(row), flush=True)
ops.gemm_tile = 0
print("per chunk: " + " | ".join(f"{NAMES[c]} {tot[c]:.3f} ms" for c in codes) + "   (x 4 chunks per 240 s track)")
