"""filtfilt / resample on a 4-minute track."""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
from scipy import signal
dev = torch.device("cuda:0")
b, a = signal.butter(N=5, Wn=48, btype="high", fs=16000)
x = torch.randn(3840000, dtype=torch.float64, device=dev) * 0.1
for blk in (8192, 2048, 512):
    for _ in range(2): ops.filtfilt_f64(x, b, a, block=blk)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): y = ops.filtfilt_f64(x, b, a, block=blk)
    torch.cuda.synchronize(); print("filtfilt block", blk, (time.perf_counter() - t0) / 5 * 1e3, "ms")
ref = signal.filtfilt(b, a, x.cpu().numpy()); print("max diff vs scipy", np.abs(y.cpu().numpy() - ref).max())
s = torch.randn(2, 10584000, device=dev)
for _ in range(2): ops.resample_poly_mono(s, 44100, 16000)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): ops.resample_poly_mono(s, 44100, 16000)
torch.cuda.synchronize(); print("resample 240 s stereo", (time.perf_counter() - t0) / 5 * 1e3, "ms")
