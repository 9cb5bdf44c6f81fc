"""One 1-D conv shape from the command line: ci co k d T [groups stride]; A/B through the AICG_CONV_* env switches."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
ci, co, k, d, T = (int(v) for v in sys.argv[1:6])
groups = int(sys.argv[6]) if len(sys.argv) > 6 else 1
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
zero = os.environ.get("KBENCH_ZERO") == "1"   # zero operands: same instruction stream, no data toggling (power-limit probe)
x = torch.randn(1, ci, T, device=dev) * (0.0 if zero else 1.0)
w = torch.randn(co, ci // groups, k) * (0.0 if zero else 0.05)
pc = ops.PackedConv(w, torch.randn(co), padding=(k - 1) * d // 2, dilation=d, stride=stride, groups=groups, device=dev)
out = torch.empty(1, co, pc.out_hw(1, T)[1], device=dev)
for _ in range(3): ops.conv(x, pc, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.conv(x, pc, out=out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
ref = torch.nn.functional.conv1d(x, w.to(dev), pc.bias, stride=stride, padding=(k - 1) * d // 2, dilation=d, groups=groups)
err = ((out - ref).pow(2).sum() / ref.pow(2).sum().clamp_min(1e-30)).sqrt().item()
print(f"{sys.argv[1:]} {dict((k_, v) for k_, v in os.environ.items() if k_.startswith('AICG_'))}: {t*1e3:8.3f} ms {2.0*co*(ci//groups)*k*out.shape[-1]/t/1e12:7.1f} TF  rel err {err:.2e}", flush=True)
