"""TDF (time-distributed fully connected) GEMMs of the MDX-Net U-Net levels, batch 16: f -> f/8 -> f with BN + ReLU (+ residual)."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops, _lib  # noqa: E402
if os.environ.get("AICG_LIB"):
    _lib._use_library_for_tests(os.environ["AICG_LIB"], "hip")
dev = torch.device("cuda:0")


def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


tot_t = tot_f = 0.0
for lvl, (c, t, f) in enumerate([(48, 256, 3072), (96, 128, 1536), (144, 64, 768), (192, 32, 384), (240, 16, 192), (288, 8, 96)]):
    x = torch.randn(16, c, t, f, device=dev)
    w1 = torch.randn(f // 8, f, device=dev) * 0.02
    w2 = torch.randn(f, f // 8, device=dev) * 0.05
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    h = ops.linear_last(x, w1, None, sc, sh, act=ops.ACT_RELU)
    t1 = timeit(lambda: ops.linear_last(x, w1, None, sc, sh, act=ops.ACT_RELU))
    t2 = timeit(lambda: ops.linear_last(h, w2, None, sc, sh, act=ops.ACT_RELU, res=x))
    fl = 2.0 * 16 * c * t * f * (f // 8)
    ref = torch.relu(torch.nn.functional.linear(x[:1, :2], w1) * sc[:2, None, None] + sh[:2, None, None])
    err = ((h[:1, :2] - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()
    print(f"L{lvl} c{c} t{t} f{f}: down {t1*1e3:7.3f} ms {fl/t1/1e12:6.1f} TF | up {t2*1e3:7.3f} ms {fl/t2/1e12:6.1f} TF | rel err {err:.1e}", flush=True)
    tot_t += t1 + t2; tot_f += 2 * fl
print(f"all levels: {tot_t*1e3:.3f} ms, {tot_f/tot_t/1e12:.1f} TF")
