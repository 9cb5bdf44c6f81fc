"""BiGRU recurrence over a 4-minute track (24 608 frames): one / two / four workgroups per direction, exchange-timeout flag and
agreement."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops
T = 24608
gi = torch.randn(1536, T, device="cuda"); whh = (torch.randn(2, 256, 768, device="cuda") / 16).contiguous(); bhh = torch.randn(1536, device="cuda") * 0.1
outs = {}
for nwg in (1, 2, 4):
    ops.GRU_WORKGROUPS = nwg
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time(); o = ops.gru_bidir(gi, whh, bhh, 256, two_workgroups=nwg > 1); torch.cuda.synchronize(); dt = time.time() - t0
    outs[nwg] = o
    print("workgroups per direction", nwg, "ms", dt * 1e3, "us/step", dt / T * 1e6, "timed out" if nwg > 1 and ops.gru_timed_out() else "", flush=True)
print("max |1 - 2|", (outs[2] - outs[1]).abs().max().item(), "max |1 - 4|", (outs[4] - outs[1]).abs().max().item())
