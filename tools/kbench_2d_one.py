"""One 2-D conv shape: ci co k N H W; A/B through the AICG_* env switches (AICG_PRECISION=bf16x3 for the split kernels)."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
ci, co, k, n, h, w_ = (int(v) for v in sys.argv[1:7])
x = torch.randn(n, ci, h, w_, device=dev)
w = torch.randn(co, ci, k, k) * 0.05
pc = ops.PackedConv(w, torch.randn(co), padding=k // 2, device=dev)
out = torch.empty(n, co, h, w_, device=dev)
for _ in range(2): ops.conv(x, pc, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.conv(x, pc, out=out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e-3
ref = torch.nn.functional.conv2d(x[:1], w.to(dev), pc.bias, padding=k // 2)
err = ((out[:1] - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()
print(f"{sys.argv[1:]} {dict((k_, v) for k_, v in os.environ.items() if k_.startswith('AICG_'))}: {t*1e3:8.3f} ms {2.0*co*ci*k*k*n*h*w_/t/1e12:7.1f} TF  rel err {err:.2e}", flush=True)
