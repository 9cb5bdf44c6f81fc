"""HuBERT feature-extractor layer (512 -> 512, k = 3, stride 2) on a rank's chunks: ablation of conv_ws3 (dev library, AICG_CONV_ABLATE)."""
import os, sys, subprocess, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
NAMES = [("full", 0), ("no x loads", 256), ("no w loads", 512), ("no loads", 768), ("no MFMA loop", 8), ("no epilogue", 16), ("stride 1 (same output count)", -1)]
if "ABL" not in os.environ:
    for name, bits in NAMES:
        subprocess.run([sys.executable, __file__], env=dict(os.environ, ABL=str(bits), ABL_NAME=name, AICG_CONV_ABLATE=str(max(bits, 0))))
    sys.exit(0)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
stride = 1 if os.environ["ABL"] == "-1" else 2
t_out = 121119
x = torch.randn(1, 512, t_out * stride + 1, device=dev)
pc = ops.PackedConv(torch.randn(512, 512, 3) * 0.03, None, stride=stride, device=dev)
out = ops.conv(x, pc, act=ops.ACT_GELU)
for _ in range(3): ops.conv(x, pc, out=out, act=ops.ACT_GELU)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.conv(x, pc, out=out, act=ops.ACT_GELU)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{os.environ['ABL_NAME']:32s} {ms:7.3f} ms {2.0*512*512*3*out.shape[-1]/ms/1e9:6.1f} TF  (T_out {out.shape[-1]})", flush=True)
