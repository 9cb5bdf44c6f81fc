"""Where the time of the F(2 x 2, 3 x 3) kernel goes (dev library: AICG_CONV_ABLATE bits of conv_w2d.h), per MDX level."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
c, t, f = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (48, 256, 3072)))
NAMES = [("full", 0), ("builtin MFMAs (untied)", 16384), ("timeline", 256), ("full + clock", 64), ("no epilogue", 16), ("epilogue without stores", 128), ("no DMA", 1), ("no fragment reads", 2),
         ("no patch reads / transform", 4), ("no MFMA", 8), ("no barriers (wrong results)", 32), ("no stage-end DMA wait (wrong results)", 32768),
         ("MFMA + fragment reads only", 1 | 4 | 16 | 32), ("MFMA only", 1 | 2 | 4 | 16 | 32), 
         ("no DMA, no epilogue", 1 | 16), ("no LDS reads at all (fragments, patch)", 2 | 4)]
if os.environ.get("KB_ONLY"):
    NAMES = [(n, b) for n, b in NAMES if str(b) in os.environ["KB_ONLY"].split(",")] + [("bits " + b, int(b)) for b in os.environ["KB_ONLY"].split(",") if int(b) not in dict((y, x) for x, y in NAMES)]
if "AICG_CONV_ABLATE" not in os.environ:   # the switch is read once per process: one child per setting
    import subprocess
    for waves in (8,):   # the variants exist for the eight-wave form
        print(f"--- C{c} {t}x{f} N16, {waves} waves", flush=True)
        for name, bits in NAMES:
            subprocess.run([sys.executable, __file__] + sys.argv[1:4], env=dict(os.environ, AICG_CONV_ABLATE=str(bits), ABL_NAME=name, AICG_W2D_WAVES=str(waves)))
    sys.exit(0)
x = torch.randn(16, c, t, f, device=dev)
w = torch.randn(c, c, 3, 3, device=dev) * 0.05
pc = ops.PackedConv(w, torch.randn(c, device=dev), padding=1, device=dev)
out = torch.empty_like(x)
ops.winograd_min_positions = 1
name = os.environ["ABL_NAME"]
for _ in range(2): ops.conv(x, pc, act=ops.ACT_RELU, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4): ops.conv(x, pc, act=ops.ACT_RELU, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 4
clk = ""
if int(os.environ["AICG_CONV_ABLATE"]) & 256:
    # wave 0 of workgroup 0: total cycles, 100 MHz ticks, cycles per phase (summed in registers, written once), its items
    v = out.view(-1)[:16 + 64].tolist()
    names = ["barrier", "burst + offsets (+ open_kstep)", "k-step 0 (+ first burst / placement)", "k-step 1", "DMA wait", "epilogue"]
    items = max(int(v[8]), 1)
    nchunk = (c + 7) // 8
    clk = "  %.0f cycles, %d items of %d stages (MFMA floor per stage and SIMD: 6144); cycles per STAGE (epilogue: per item), by wave:" % (v[0], items, nchunk)
    for w in range(8):
        ph = v[16 + 8 * w:16 + 8 * w + 6]
        clk += "\n      wave %d: " % w + ", ".join("%s %.0f" % (n, x / items / (nchunk if i < 5 else 1)) for i, (n, x) in enumerate(zip(names, ph)))
if int(os.environ["AICG_CONV_ABLATE"]) & 64:
    cyc, ticks = out.view(-1)[:2].tolist()   # s_memtime cycles, 100 MHz ticks of workgroup 0
    clk = f"  workgroup 0: {cyc:.0f} cycles in {ticks / 100:.1f} us = {cyc / max(ticks, 1) * 0.1:.3f} GHz"
print(f"{name:45s} {ms:7.3f} ms{clk}", flush=True)
