"""GPU-box timing of the synthesizer on one reference-sized chunk (not the judged bench)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aicovergen_amd.infer_pack.models import SynthesizerTrnMs768NSFsid
from synthetic import weights
from synthetic.inputs import synth_inputs
cfg = weights.SYNTH_CFG_40K_V2
net = SynthesizerTrnMs768NSFsid(*cfg, is_half=False); del net.enc_q
net.load_state_dict(weights.synth_state_dict(cfg, 1234), strict=False); net.eval().to("cuda:0")
res = {}
for T in (2000, 6600):
    phone, pitch, f0, nz, ns = synth_inputs(cfg, T, 5)
    phone, pitch, f0, nz, ns = [t.cuda() for t in (phone, pitch, f0, nz, ns)]
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        o, _, _ = net.infer(phone, torch.tensor([T]), pitch, f0, torch.tensor([1]), noise_z=nz, noise_src=ns)
        torch.cuda.synchronize(); dt = time.time() - t0
    audio_s = T / 100.0
    res["synth_T%d" % T] = {"s": dt, "rtf": audio_s / dt, "tflops": 96.85e9 * audio_s / dt / 1e12,
                            "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30}
    print(res, flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "kbench_synth.json"), "w"), indent=1)
