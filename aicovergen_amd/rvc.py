"""Model construction / file-level driver of the RVC stage behind the reference's src/rvc.py surface:
Config, load_hubert, get_vc, rvc_infer (same signatures and return values).  fairseq is not needed: the HuBERT
checkpoint is read directly (aicovergen_amd.hubert.load_state)."""
from multiprocessing import cpu_count
from pathlib import Path

import numpy as np
import torch
from scipy.io import wavfile

from .hubert import HubertModel, load_state
from .infer_pack.models import (SynthesizerTrnMs256NSFsid, SynthesizerTrnMs256NSFsid_nono, SynthesizerTrnMs768NSFsid,
                                SynthesizerTrnMs768NSFsid_nono)
from .my_utils import load_audio
from .vc_infer_pipeline import VC

BASE_DIR = Path(__file__).resolve().parent.parent


class Config:
    """Chunk-geometry presets of the reference (src/rvc.py:20-95).  The NVIDIA-name heuristics that rewrite files
    under src/ for 16-series / P40 cards (:33-50) do not apply to an MI355X and are intentionally not carried over."""

    def __init__(self, device, is_half):
        self.device = device
        self.is_half = is_half
        self.n_cpu = 0
        self.gpu_name = None
        self.gpu_mem = None
        self.x_pad, self.x_query, self.x_center, self.x_max = self.device_config()

    def device_config(self) -> tuple:
        if torch.cuda.is_available():
            i_device = int(str(self.device).split(":")[-1]) if ":" in str(self.device) else 0
            self.gpu_mem = int(torch.cuda.get_device_properties(i_device).total_memory / 1024 / 1024 / 1024 + 0.4)
        else:
            print("No supported N-card found, use CPU for inference")
            self.device = "cpu"
            self.is_half = True
        if self.n_cpu == 0:
            self.n_cpu = cpu_count()
        if self.is_half:
            x_pad, x_query, x_center, x_max = 3, 10, 60, 65   # "6G memory config"
        else:
            x_pad, x_query, x_center, x_max = 1, 6, 38, 41    # "5G memory config"
        if self.gpu_mem is not None and self.gpu_mem <= 4:
            x_pad, x_query, x_center, x_max = 1, 5, 30, 32
        return x_pad, x_query, x_center, x_max


def load_hubert(device, is_half, model_path):
    hubert = HubertModel(load_state(model_path))
    hubert = hubert.to(device)
    hubert = hubert.half() if is_half else hubert.float()
    hubert.eval()
    return hubert


def get_vc(device, is_half, config, model_path):
    cpt = torch.load(model_path, map_location='cpu')
    if "config" not in cpt or "weight" not in cpt:
        raise ValueError(f'Incorrect format for {model_path}. Use a voice model trained using RVC v2 instead.')
    tgt_sr = cpt["config"][-1]
    cpt["config"][-3] = cpt["weight"]["emb_g.weight"].shape[0]
    if_f0 = cpt.get("f0", 1)
    version = cpt.get("version", "v1")
    if version == "v1":
        net_g = SynthesizerTrnMs256NSFsid(*cpt["config"], is_half=is_half) if if_f0 == 1 else SynthesizerTrnMs256NSFsid_nono(*cpt["config"])
    elif version == "v2":
        net_g = SynthesizerTrnMs768NSFsid(*cpt["config"], is_half=is_half) if if_f0 == 1 else SynthesizerTrnMs768NSFsid_nono(*cpt["config"])
    del net_g.enc_q
    print(net_g.load_state_dict(cpt["weight"], strict=False))
    net_g.eval().to(device)
    net_g = net_g.half() if is_half else net_g.float()
    vc = VC(tgt_sr, config)
    return cpt, version, net_g, tgt_sr, vc


def rvc_infer(index_path, index_rate, input_path, output_path, pitch_change, f0_method, cpt, version, net_g, filter_radius,
              tgt_sr, rms_mix_rate, protect, crepe_hop_length, vc, hubert_model):
    audio = load_audio(input_path, 16000)
    times = [0, 0, 0]
    if_f0 = cpt.get('f0', 1)
    audio_opt = vc.pipeline(hubert_model, net_g, 0, audio, input_path, times, pitch_change, f0_method, index_path, index_rate,
                            if_f0, filter_radius, tgt_sr, 0, rms_mix_rate, version, protect, crepe_hop_length)
    wavfile.write(output_path, tgt_sr, audio_opt)
