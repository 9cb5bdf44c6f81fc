"""TDF block of MDX-Net levels 0-2 (batch 16): the fused kernel (aicg_tdf_pair) against the two NT GEMMs it replaces."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")


def timeit(fn, iters=4, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for lvl, (c, t, f) in enumerate([(48, 256, 3072), (96, 128, 1536), (144, 64, 768)]):
    x = torch.randn(16, c, t, f, device=dev)
    h = f // 8
    w1 = torch.randn(h, f, device=dev) * 0.02
    w2 = torch.randn(f, h, device=dev) * 0.05
    b1, b2 = torch.randn(h, device=dev) * 0.1, torch.randn(f, device=dev) * 0.1
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    w1p, w2p = ops.pack_tdf_w1(w1), ops.pack_tdf_w2(w2)
    out = torch.empty_like(x)
    fl = 4.0 * 16 * c * t * f * h
    two = lambda: ops.linear_last(ops.linear_last(x, w1, b1, sc, sh, act=ops.ACT_RELU), w2, b2, sc, sh, act=ops.ACT_RELU, res=x)
    ref = two()
    t2 = timeit(two)
    line = f"L{lvl} c{c} t{t} f{f}: two GEMMs {t2*1e3:7.3f} ms {fl/t2/1e12:6.1f} TF |"
    fn = lambda: ops.tdf_pair(x, w1p, b1, sc, sh, w2p, b2, sc, sh, out=out)
    tf = timeit(fn)
    err = ((out - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()
    line += f" fused: {tf*1e3:6.3f} ms {fl/tf/1e12:5.1f} TF (rel diff {err:.0e}) |"
    print(line, flush=True)
