"""RMVPE's deep 3x3 layers (few positions, long K) under the dispatcher's fill target AICG_CONV_WANT (dev library; default 512 workgroups:
below it the tile shrinks).  One child per setting (the switch is read once per process)."""
import os, sys, subprocess, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if "WANT_CHILD" not in os.environ:
    for want in (sys.argv[1:] or ["512", "256", "128", "64"]):
        subprocess.run([sys.executable, __file__], env=dict(os.environ, WANT_CHILD=want, AICG_CONV_WANT=want))
    sys.exit(0)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
line = f"want {os.environ['WANT_CHILD']:>4s}"
with ops.fp32_layers():
    for c, h, w in [(512, 769, 4), (256, 1538, 8), (128, 3076, 16), (64, 6152, 32), (32, 12304, 64), (16, 24608, 128), (512, 385, 4), (256, 769, 8)]:
        x = torch.randn(1, c, h, w, device=dev)
        pc = ops.PackedConv(torch.randn(c, c, 3, 3) * 0.03, torch.randn(c), padding=1, device=dev)
        out = torch.empty_like(x)
        for _ in range(3): ops.conv(x, pc, out=out, act=ops.ACT_RELU, res=x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv(x, pc, out=out, act=ops.ACT_RELU, res=x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f" | C{c} {h}x{w} {ms*1e3:6.1f} us {2.0*c*c*9*h*w/ms/1e9:5.1f}"
    # HuBERT / enc_p-sized GEMMs that the same rule tiles
    for ci, co, k, t in [(192, 384, 5, 6420), (192, 192, 1, 6420), (768, 192, 1, 6420), (192, 768, 3, 6420), (768, 192, 3, 6420)]:
        x = torch.randn(1, ci, t, device=dev)
        pc = ops.PackedConv(torch.randn(co, ci, k) * 0.03, torch.randn(co), padding=k // 2, device=dev)
        out = torch.empty(1, co, t, device=dev)
        for _ in range(3): ops.conv(x, pc, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv(x, pc, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f" | {ci}>{co}k{k}@{t} {ms*1e3:6.1f} us {2.0*ci*co*k*t/ms/1e9:5.1f}"
print(line, flush=True)
