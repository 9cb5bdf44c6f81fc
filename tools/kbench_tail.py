"""Tile quantisation of conv_g1w launches: the same layer at lengths that are a whole number of rounds of 512 workgroup slots and just above."""
import os, sys, statistics, torch
os.environ.setdefault("AICG_DEV", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")


def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 2)
    return statistics.median(ts)


for c, k in ((128, 11), (128, 3), (64, 7), (256, 7)):
    pc = ops.PackedConv(torch.randn(c, c, k, device=dev) * 0.05, torch.randn(c, device=dev) * 0.1, padding=(k - 1) // 2, device=dev)
    rows = c // 32
    for rounds in (2, 10):
        base = rounds * 512 // rows * 512                 # positions of `rounds` whole rounds of 512 slots
        for T in (base, base + 512 * 8 // rows, base + 512 * 64 // rows, base + 512 * 256 // rows):
            x = torch.randn(1, c, T, device=dev)
            out = torch.empty_like(x)
            ms = bench(lambda: ops.conv(x, pc, res=x, pre_act=ops.ACT_LRELU, pre_slope=0.1, out=out))
            tiles = rows * ((T + 511) // 512)
            print(f"C{c} k{k}: T {T:8d} = {tiles:5d} tiles = {tiles / 512:6.3f} rounds: {ms * 1e3:8.1f} us, {ms * 1e6 / tiles * 512:8.1f} us per round of work, {2.0 * c * c * k * T / ms / 1e9:6.1f} TF direct", flush=True)
