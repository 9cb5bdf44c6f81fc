"""Attention variants A/B (dev library, AICG_ATTN_VAR of attn.hip): HuBERT (12 x 64, T = 3300) and enc_p (2 x 96, T = 6600, window 10).
The switch is read once per process: one child per (variant, round); rounds interleave the variants (the shader clock sags over a run)."""
import os, sys, subprocess, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
VARS = [("r3 registers (2 / 1 waves per SIMD)", 8), ("3 / 2 waves per SIMD", 0), ("+ exp2", 1), ("+ lazy rescale", 2), ("+ both", 3),
        ("4 / 3 waves per SIMD (spills)", 4), ("4 / 3 waves + both", 7)]
if "ATTN_CHILD" not in os.environ:
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for r in range(rounds):
        for name, v in VARS:
            subprocess.run([sys.executable, __file__], env=dict(os.environ, ATTN_CHILD=name, AICG_ATTN_VAR=str(v)))
    sys.exit(0)
from aicovergen_amd import _lib, ops  # noqa: E402
_lib._use_library_for_tests(os.path.join(ROOT, "aicovergen_amd", "libaicg_hip_dev.so"), "hip")
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


torch.manual_seed(0)
line = f"{os.environ['ATTN_CHILD']:40s}"
for name, H, D, T, win in [("hubert", 12, 64, 3300, 0), ("enc_p", 2, 96, 6600, 10)]:
    q, k, v = (torch.randn(H * D, T, device=dev) * 0.3 for _ in range(3))
    relk = torch.randn(H, 2 * win + 1, T, device=dev) * 0.1 if win else None
    ev = torch.randn(2 * win + 1, D, device=dev) * 0.1 if win else None
    # reference: dense softmax in float64 on a slice of the queries
    qs = slice(1000, 1128)
    qq, kk, vv = (a.double().view(H, D, T) for a in (q, k, v))
    s = torch.einsum("hdi,hdj->hij", qq[:, :, qs] * 0.125, kk)
    if win:
        i = torch.arange(1000, 1128, device=dev)[:, None]; j = torch.arange(T, device=dev)[None, :]
        dl = j - i
        band = (dl.abs() <= win)
        rk = relk.double()   # (H, 2w+1, T): q_i . E_m at query i
        add = torch.zeros(H, 128, T, device=dev, dtype=torch.float64)
        for h in range(H):
            add[h][band] = rk[h][(dl + win).clamp(0, 2 * win)[band], i.expand_as(dl)[band]]
        s = s + add
    pr = torch.softmax(s, -1)
    ref = torch.einsum("hij,hdj->hdi", pr, vv)
    if win:
        for h in range(H):
            pb = torch.zeros(128, 2 * win + 1, device=dev, dtype=torch.float64)
            for m in range(2 * win + 1):
                jj = torch.arange(1000, 1128, device=dev) + m - win
                pb[:, m] = pr[h][torch.arange(128, device=dev), jj]
            ref[h] += (pb @ ev.double()).T
    for sp in ((None, 2, 3, 6) if name == "hubert" else (None, 8)):
        fn = lambda: ops.attention(q, k, v, H, relk=relk, relv_emb=ev, window=win, scale=0.125, n_splits=sp)
        o = fn().view(H, D, T)[:, :, qs].double()
        err = ((o - ref).pow(2).mean() / ref.pow(2).mean()).sqrt().item()
        t = timeit(fn)
        line += f" | {name} s={sp}: {t*1e3:6.3f} ms {4.0*T*T*H*D/t/1e12:5.1f} TF err {err:.1e}"
print(line, flush=True)
