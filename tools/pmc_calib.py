"""FETCH_SIZE / WRITE_SIZE calibration on known byte counts (run under `rocprofv3 --pmc FETCH_SIZE` and again with WRITE_SIZE):
  copy     torch clone of a 1.2 GB fp32 tensor (16 B/lane streaming reads)                 -> reads 1.2 GB, writes 1.2 GB
  conv_L1  MDX level-1 conv, 16 x 96 x 128 x 1536 in and out (4 B/lane patch loads + 16 B/lane weight loads), halo re-reads served by L2
  tdf      NT-GEMM 196608 x 3072 -> 384 (16 B/lane loads of A and W)
Prints the algorithmic bytes per call; tools/pmc_calib_report.py divides the counters by them."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
x = torch.randn(16, 96, 128, 1536, device=dev)
for _ in range(3):
    y = x.clone()
torch.cuda.synchronize()
pc = ops.PackedConv(torch.randn(96, 96, 3, 3) * 0.05, torch.zeros(96), padding=1, device=dev)
out = torch.empty_like(x)
for _ in range(3):
    ops.conv(x, pc, out=out, act=ops.ACT_RELU)
torch.cuda.synchronize()
a = torch.randn(16, 48, 256, 3072, device=dev)
w1 = torch.randn(384, 3072, device=dev) * 0.02
for _ in range(3):
    ops.linear_last(a, w1, None, None, None, act=ops.ACT_RELU)
torch.cuda.synchronize()
print("algorithmic bytes: copy read %d write %d | conv_L1 read %d write %d | tdf read %d write %d" % (
    x.numel() * 4, x.numel() * 4, x.numel() * 4 + 96 * 96 * 9 * 4, x.numel() * 4, a.numel() * 4 + w1.numel() * 4, a.numel() // 8 * 4))
