"""GPU-box timing of mangio-crepe f0 on a 4-min track (config 4 of BASELINE.json), not the judged bench."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicovergen_amd import crepe
from synthetic import weights
from synthetic.inputs import vocal_like
net = crepe.Crepe(weights.crepe_state_dict(weights.CREPE_FULL, 1234), "cuda:0")
for secs in (30.0, 246.0):
    a = vocal_like(secs, 16000, 1)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        f0 = crepe.mangio_crepe_f0(net, a, int(secs * 100), 128)
        torch.cuda.synchronize(); dt = time.time() - t0
    frames = 1 + len(a) // 128
    print(secs, "s audio:", dt, "s; frames", frames, "TFLOP/s", 2 * 1.41e9 * frames / dt / 1e12, flush=True)
