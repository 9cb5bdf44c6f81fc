"""One MDX level through ONE form of the F(2 x 2, 3 x 3) kernel, a few launches -- the command rocprofv3 --pmc wraps."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
c, t, f = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(16, c, t, f, device=dev)
w = torch.randn(c, c, 3, 3, device=dev) * 0.05
pc = ops.PackedConv(w, torch.randn(c, device=dev), padding=1, device=dev)
out = torch.empty_like(x)
ops.winograd_min_positions = 1
for _ in range(4): ops.conv(x, pc, act=ops.ACT_RELU, out=out)
torch.cuda.synchronize()
