"""Where a Winograd stage's time goes (dev library: AICG_CONV_ABLATE bits of conv_ws3w.h) on the MDX level-1 TFC layer."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
c, t, f = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (96, 128, 1536)))
x = torch.randn(16, c, t, f, device=dev)
w = torch.randn(c, c, 3, 3, device=dev) * 0.05
pc = ops.PackedConv(w, torch.randn(c, device=dev), padding=1, device=dev)
out = torch.empty_like(x)
ops.winograd_min_positions = 1
NAMES = [("full", 0), ("full + clock", 64), ("MFMA only + clock", 126), ("no x loads + clock", 68), ("x loads replaced by register moves (transform + commit stay)", 128), ("x loads issued, results unused (no transform)", 256), ("dword patch loads on interior tiles too", 512), ("no MFMA / LDS reads", 1), ("MFMA on one LDS address", 2), ("no x loads", 4), ("no w loads", 32), ("no loads", 36),
                   ("no commit", 8), ("no epilogue", 16), ("no loads, no commit", 44), ("MFMA only (no loads/commit/epilogue, one address)", 62),
                   ("producer only", 17)]
if "AICG_CONV_ABLATE" not in os.environ:   # the switch is read once per process: one child per setting
    import subprocess
    for name, bits in NAMES:
        subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, AICG_CONV_ABLATE=str(bits), ABL_NAME=name))
    sys.exit(0)
for name in [os.environ["ABL_NAME"]]:
    for _ in range(2): ops.conv(x, pc, act=ops.ACT_RELU, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): ops.conv(x, pc, act=ops.ACT_RELU, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    clk = ""
    if int(os.environ["AICG_CONV_ABLATE"]) & 64:
        cyc, ticks = out.view(-1)[:2].tolist()   # s_memtime cycles, 100 MHz ticks of workgroup 0
        clk = f"  workgroup 0: {cyc:.0f} cycles in {ticks / 100:.1f} us = {cyc / max(ticks, 1) * 0.1:.3f} GHz"
    print(f"{name:55s} {ms:7.3f} ms{clk}", flush=True)
