"""One conv shape, a few launches: target for rocprofv3 --pmc runs.  usage: kbench_one.py n ci co H W k [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402

n, ci, co, H, W, k = [int(v) for v in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 3
dev = torch.device("cuda:0")
x = torch.randn(n, ci, H, W, device=dev)
pc = ops.PackedConv(torch.randn(co, ci, k, k) * 0.05, torch.randn(co), padding=k // 2, device=dev)
out = torch.empty(n, co, H, W, device=dev)
for _ in range(iters):
    ops.conv(x, pc, out=out, act=ops.ACT_RELU)
torch.cuda.synchronize()
