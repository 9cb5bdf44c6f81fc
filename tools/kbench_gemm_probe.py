"""HuBERT's per-token GEMMs (1 x 1 convolutions over a (C, T) map): what the rate depends on -- row alignment of T, tile height,
fragment form (dev library switches; one child per setting, the switches are read once per process)."""
import os, sys, subprocess, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
SETS = [("default T=13198", {}, 13198), ("T=13200 (rows 16-byte aligned)", {}, 13200), ("T=13312 (= 104 x 128)", {}, 13312),
        ("64-row tiles", {"AICG_CONV_FORCE_BM": "64"}, 13198), ("4-byte fragments (conv_ws)", {"AICG_CONV_V3": "0"}, 13198),
        ("single-role kernels", {"AICG_CONV_WS": "0"}, 13198), ("T=6599", {}, 6599), ("T=26396", {}, 26396)]
if "GEMM_CHILD" not in os.environ:
    for name, env, t in SETS:
        subprocess.run([sys.executable, __file__], env=dict(os.environ, GEMM_CHILD=name, GEMM_T=str(t), **env))
    sys.exit(0)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")
t = int(os.environ["GEMM_T"])
line = f"{os.environ['GEMM_CHILD']:34s}"
for ci, co, act in [(768, 3072, ops.ACT_GELU), (768, 3072, ops.ACT_NONE), (3072, 768, ops.ACT_NONE), (768, 768, ops.ACT_NONE), (768, 1536, ops.ACT_NONE)]:
    x = torch.randn(1, ci, t, device=dev)
    pc = ops.PackedConv(torch.randn(co, ci, 1) * 0.03, torch.randn(co), device=dev)
    out = torch.empty(1, co, t, device=dev)
    for _ in range(3): ops.conv(x, pc, out=out, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv(x, pc, out=out, act=act)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    line += f" | {ci}>{co}{'g' if act == ops.ACT_GELU else ' '} {ms*1e3:6.1f} us {2.0*ci*co*t/ms/1e9:5.1f}"
print(line, flush=True)
