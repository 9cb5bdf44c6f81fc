"""HuBERT's per-token GEMMs as 1 x 1 convolutions over the (C, T) map of a rank's chunks (T = 13198): tile choice A/B (dev library)."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
if os.environ.get("AICG_LIB"):
    _lib._use_library_for_tests(os.environ["AICG_LIB"], "hip")
dev = torch.device("cuda:0")
for ci, co, act in [(768, 3072, ops.ACT_GELU), (3072, 768, ops.ACT_NONE), (768, 768, ops.ACT_NONE), (768, 1536, ops.ACT_NONE), (512, 768, ops.ACT_NONE)]:
    t = 13198
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
    print(f"C{ci}>{co} T{t}: {ms*1e3:7.1f} us {2.0*ci*co*t/ms/1e9:6.1f} TF", flush=True)
