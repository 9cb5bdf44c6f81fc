"""tdf_pair<12> (MDX level 0: F = 3072, H = 384, batch 16) with one of the kernel's profiling variants (development library,
AICG_TDF_ABLATE = bits: 1 no residual loads, 2 no output stores, 4 no phase-2 weight DMA, 8 no phase-1 x loads, 16 no phase-1 weight DMA,
32 no phase-2 MFMAs, 64 no phase-1 MFMAs).  One variant per process (the switch is read once):
    for a in 0 1 2 3 4 7 8 16 24 31 32 64 96 0; do AICG_TDF_ABLATE=$a python tools/kbench_tdf_ablate.py; done"""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
c, t, f = 48, 256, 3072
h = f // 8
x = torch.randn(16, c, t, f, device=dev)
w1, w2 = torch.randn(h, f, device=dev) * 0.02, torch.randn(f, h, device=dev) * 0.05
b1, b2 = torch.randn(h, device=dev) * 0.1, torch.randn(f, device=dev) * 0.1
sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
w1p, w2p = ops.pack_tdf_w1(w1), ops.pack_tdf_w2(w2)
out = torch.empty_like(x)
fn = lambda: ops.tdf_pair(x, w1p, b1, sc, sh, w2p, b2, sc, sh, out=out)
for _ in range(3): fn()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 3)
ts.sort()
fl = 4.0 * 16 * c * t * f * h
print("AICG_TDF_ABLATE=%-3s  %.3f ms (min %.3f)  %.1f TFLOP/s if it were the whole block" % (os.environ.get("AICG_TDF_ABLATE", "0"), ts[2], ts[0], fl / ts[2] / 1e9), flush=True)
