"""Measured fp32-MFMA ceiling of this device (pure issue loop, random-ish operands)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import _lib
out = torch.empty(2048 * 256, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for nb in (256, 512, 1024, 2048):
    iters = 20000
    _lib.call("aicg_mfma_probe", out.data_ptr(), nb, 100, 0.37, st)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.call("aicg_mfma_probe", out.data_ptr(), nb, iters, 0.37, st); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    fl = nb * 4 * iters * 4 * 2.0 * 32 * 32 * 2
    print("blocks %d: %.1f TFLOP/s" % (nb, fl / t / 1e12), flush=True)
