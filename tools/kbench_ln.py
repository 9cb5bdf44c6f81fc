"""LayerNorm over channels of (1, 768, T): the batched-HuBERT shape (T = 13198) and the per-chunk one (T = 3300)."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib, ops  # noqa: E402
if os.environ.get("AICG_LIB"):
    _lib._use_library_for_tests(os.environ["AICG_LIB"], "hip")
dev = torch.device("cuda:0")
for c, t in [(768, 13198), (768, 3300), (192, 6600), (768, 26400)]:
    x = torch.randn(1, c, t, device=dev); r = torch.randn(1, c, t, device=dev)
    g = torch.randn(c, device=dev); b = torch.randn(c, device=dev); out = torch.empty_like(x)
    for _ in range(3): ops.layernorm_ct(x, g, b, res=r, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.layernorm_ct(x, g, b, res=r, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"C{c} T{t}: {us:7.1f} us  {3 * x.numel() * 4 / us / 1e3:7.1f} GB/s", flush=True)
