"""The vocoder's 256-channel ResBlock layers (73 080 positions: 2 x 571 tiles of 128 x 128 = 2.2 rounds over 512 slots) under a forced tile
height (dev library, AICG_CONV_FORCE_BM; one child per setting)."""
import os, sys, subprocess, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if "BM_CHILD" not in os.environ:
    for bm in (sys.argv[1:] or ["0", "64", "96", "0"]):
        subprocess.run([sys.executable, __file__], env=dict(os.environ, BM_CHILD=bm, AICG_CONV_FORCE_BM=bm))
    sys.exit(0)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
line = f"BM {os.environ['BM_CHILD']:>3s}"
for c, t in [(256, 73080), (256, 64200), (128, 730800)]:
    for k, d in [(3, 1), (7, 3), (11, 5), (11, 1)]:
        x = torch.randn(1, c, t, device=dev)
        pc = ops.PackedConv(torch.randn(c, c, k) * 0.03, torch.randn(c), padding=(k - 1) * d // 2, dilation=d, device=dev)
        out = torch.empty_like(x)
        fn = lambda: ops.conv(x, pc, out=out, res=x, pre_act=ops.ACT_LRELU, pre_slope=0.1)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f" | C{c}@{t} k{k}d{d} {ms*1e3:6.1f} us {2.0*c*c*k*t/ms/1e9:5.1f}"
print(line, flush=True)
