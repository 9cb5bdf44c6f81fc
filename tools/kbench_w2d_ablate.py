"""Where the time of the F(2 x 2, 3 x 3) kernel goes (dev library: AICG_CONV_ABLATE bits of conv_w2d.h), per MDX level."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
c, t, f = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (48, 256, 3072)))
NAMES = [("full", 0), ("builtin MFMAs (untied)", 16384), ("timeline", 256), ("full + clock", 64), ("no epilogue", 16), ("epilogue without stores", 128), ("no DMA", 1), ("no fragment reads", 2),
         ("no patch reads / transform", 4), ("no MFMA", 8), ("no barriers (wrong results)", 32),
         ("MFMA + fragment reads only", 1 | 4 | 16 | 32), ("MFMA only", 1 | 2 | 4 | 16 | 32), ("MFMA only + clock", 1 | 2 | 4 | 16 | 32 | 64),
         ("DMA + barriers only", 2 | 4 | 8 | 16), ("no DMA, no epilogue", 1 | 16), ("no LDS reads at all (fragments, patch)", 2 | 4)]
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
    # wave 0 of workgroup 0: per stage [entry, after barrier, after DMA issue / lane offsets, after k-step 0, after k-step 1, after DMA wait],
    # per item two more around the epilogue's body
    v = out.view(-1)[:8192].tolist()
    n = int(v[1])
    st = v[2:n]
    nchunk = (c + 7) // 8
    per_item = 6 * nchunk + 2
    items = len(st) // per_item
    import statistics as S
    acc = {"barrier wait": [], "issue": [], "kstep0": [], "kstep1": [], "dma wait": [], "stage": [], "pre-epilogue": [], "epilogue": [], "item": []}
    for it in range(1, items - 1):
        b = st[it * per_item:(it + 1) * per_item]
        for g in range(nchunk):
            e = b[6 * g:6 * g + 6]
            acc["barrier wait"].append(e[1] - e[0]); acc["issue"].append(e[2] - e[1]); acc["kstep0"].append(e[3] - e[2])
            acc["kstep1"].append(e[4] - e[3]); acc["dma wait"].append(e[5] - e[4]); acc["stage"].append(e[5] - e[0])
        acc["pre-epilogue"].append(b[6 * nchunk] - b[6 * nchunk - 1]); acc["epilogue"].append(b[6 * nchunk + 1] - b[6 * nchunk])
        acc["item"].append(st[(it + 1) * per_item] - b[0])
    clk = "  cycles (median / mean over %d items): " % (items - 2) + ", ".join("%s %.0f / %.0f" % (k, S.median(x), S.mean(x)) for k, x in acc.items() if x)
    first = [st[per_item + 6 * g + 5] - st[per_item + 6 * g] for g in range(nchunk)]
    clk += "\n      stages of item 1: " + " ".join("%.0f" % x for x in first)
if int(os.environ["AICG_CONV_ABLATE"]) & 64:
    cyc, ticks = out.view(-1)[:2].tolist()   # s_memtime cycles, 100 MHz ticks of workgroup 0
    clk = f"  workgroup 0: {cyc:.0f} cycles in {ticks / 100:.1f} us = {cyc / max(ticks, 1) * 0.1:.3f} GHz"
print(f"{name:45s} {ms:7.3f} ms{clk}", flush=True)
