"""Attention micro-benchmark on the GPU box: HuBERT (12 x 64, T = 3300) and enc_p (2 x 96, T = 6600, window 10) shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for name, H, D, T, win in [("hubert", 12, 64, 3300, 0), ("enc_p", 2, 96, 6600, 10)]:
    q, k, v = (torch.randn(H * D, T, device=dev) * 0.3 for _ in range(3))
    relk = torch.randn(H, 2 * win + 1, T, device=dev) * 0.1 if win else None
    ev = torch.randn(2 * win + 1, D, device=dev) * 0.1 if win else None
    for s in (1, 2, 4, 8, None):
        t = timeit(lambda: ops.attention(q, k, v, H, relk=relk, relv_emb=ev, window=win, n_splits=s))
        print(f"{name:8s} splits={s}: {t*1e3:7.3f} ms  {4.0*T*T*H*D/t/1e12:6.1f} TF", flush=True)
