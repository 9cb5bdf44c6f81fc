"""A/B of conv tile variants on vocoder shapes (AICG_CONV_8WAVE=0/1 set by the caller)."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e-3
for (c, L) in [(256, 66000), (128, 660000)]:
    for k in (3, 7, 11):
        x = torch.randn(1, c, L, device=dev); pc = ops.PackedConv(torch.randn(c, c, k) * 0.05, torch.randn(c), padding=(k - 1) // 2, device=dev)
        out = torch.empty_like(x)
        t = timeit(lambda: ops.conv(x, pc, res=x, out=out, pre_act=ops.ACT_LRELU, pre_slope=0.1))
        print("8wave=%s c%d k%d: %.3f ms %.1f TF" % (os.environ.get("AICG_CONV_8WAVE", "1"), c, k, t * 1e3, 2.0 * c * c * k * L / t / 1e12), flush=True)
x = torch.randn(16, 240, 16, 192, device=dev); pc = ops.PackedConv(torch.randn(240, 240, 3, 3) * 0.05, torch.randn(240), padding=1, device=dev); out = torch.empty_like(x)
t = timeit(lambda: ops.conv(x, pc, out=out, act=ops.ACT_RELU)); print("mdx_L4_c240 %.3f ms %.1f TF" % (t * 1e3, 2.0 * 16 * 240 * 240 * 9 * 16 * 192 / t / 1e12))
