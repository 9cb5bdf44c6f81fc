"""Time one 2-D conv shape: kbench_case.py n ci co H W k"""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
n, ci, co, H, W, k = [int(v) for v in sys.argv[1:7]]
dev = torch.device("cuda:0")
x = torch.randn(n, ci, H, W, device=dev)
pc = ops.PackedConv(torch.randn(co, ci, k, k) * 0.05, torch.randn(co), padding=k // 2, device=dev)
out = torch.empty(n, co, H, W, device=dev)
for _ in range(3): ops.conv(x, pc, out=out, act=ops.ACT_RELU)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.conv(x, pc, out=out, act=ops.ACT_RELU)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print(os.environ.get("AICG_CONV_ABLATE", "0"), f"{t*1e3:8.3f} ms {2.0*n*co*ci*k*k*H*W/t/1e12:7.1f} TF(nominal)", flush=True)
