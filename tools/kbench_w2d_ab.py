"""A/B of the conv_w2d forms (aicg_conv_desc.wino codes: 2 / 3 eight / four waves, 4 / 5 the same on quad fragments; 1 = the row form) on the MDX levels, ROUND-ROBIN: the clock of a hot chip sags over a
sustained run, so forms measured one after the other are not comparable (r4: the same kernel 2.82 ms first and 3.20 ms last in one
process).  Every round times each form for a few launches; the median over the rounds is reported."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
codes = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "2,3,4,5,1")]   # 1 = row form; 6: schedule variant under test (dev library)
if any(c >= 6 for c in codes):
    _lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
rounds = int(os.environ.get("KB_ROUNDS", "7"))
ops.winograd_min_positions = 1


def run(code, x, pc, out):
    ops.winograd2d = code != 1
    ops.winograd2d_code = code if code > 1 else 0
    ops.conv(x, pc, act=ops.ACT_RELU, out=out)


for lvl, (c, t, f) in enumerate([(48, 256, 3072), (96, 128, 1536), (144, 64, 768), (192, 32, 384), (240, 16, 192)]):
    x = torch.randn(16, c, t, f, device=dev)
    ops.winograd2d = True          # (read at pack time as well)
    pc = ops.PackedConv(torch.randn(c, c, 3, 3, device=dev) * 0.05, torch.randn(c, device=dev) * 0.1, padding=1, device=dev)
    out = torch.empty_like(x)
    outs = {}
    for code in codes:
        for _ in range(3): run(code, x, pc, out)
        outs[code] = out.clone()
    torch.cuda.synchronize()
    diff = max(float((outs[c] - outs[codes[0]]).abs().max()) for c in codes)
    times = {code: [] for code in codes}
    for r in range(rounds):
        for code in (codes if r % 2 == 0 else codes[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): run(code, x, pc, out)
            e1.record(); torch.cuda.synchronize()
            times[code].append(e0.elapsed_time(e1) / 3)
    print(f"L{lvl} c{c}: " + " | ".join(f"code {code}: {statistics.median(v):6.3f} ms (min {min(v):6.3f})" for code, v in times.items()) + f" | max diff between forms {diff:.1e}", flush=True)
