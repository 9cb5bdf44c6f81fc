"""File-level driver of the RVC stage behind the call surface of the reference's src/rvc.py:
`Config(device, is_half)`, `load_hubert(device, is_half, model_path)`, `get_vc(device, is_half, config, model_path)` and
`rvc_infer(...)` keep the reference's argument order, return values and side effects (src/rvc.py:20-151), so
src/main.py:186-199 drives this module unchanged.  fairseq is not needed: the HuBERT checkpoint is read directly
(aicovergen_amd.hubert.load_state)."""
import multiprocessing
import pathlib

import torch
from scipy.io import wavfile

from . import hubert as _hubert
from .infer_pack import models as _models
from .my_utils import load_audio
from .vc_infer_pipeline import VC

BASE_DIR = pathlib.Path(__file__).resolve().parent.parent

# chunk geometry presets (x_pad, x_query, x_center, x_max) in seconds, src/rvc.py:76-95
_PRESET_HALF = (3, 10, 60, 65)      # the reference's "6G memory config", chosen when is_half is set
_PRESET_FULL = (1, 6, 38, 41)       # "5G memory config"
_PRESET_SMALL_CARD = (1, 5, 30, 32)  # cards with <= 4 GB

# (version, has f0) -> synthesizer class, src/rvc.py:126-132
_SYNTHESIZERS = {
    ("v1", True): _models.SynthesizerTrnMs256NSFsid,
    ("v1", False): _models.SynthesizerTrnMs256NSFsid_nono,
    ("v2", True): _models.SynthesizerTrnMs768NSFsid,
    ("v2", False): _models.SynthesizerTrnMs768NSFsid_nono,
}


class Config:
    """Device / precision / chunk-geometry holder with the reference's attribute names.  The NVIDIA-name heuristics that
    rewrite files under src/ for 16-series / P40 cards (src/rvc.py:33-50) do not apply to an MI355X and are not carried
    over; everything else follows the reference, including its quirk of switching `is_half` ON when it falls back to CPU."""

    def __init__(self, device, is_half):
        self.device, self.is_half = device, is_half
        self.n_cpu, self.gpu_name, self.gpu_mem = 0, None, None
        self.x_pad, self.x_query, self.x_center, self.x_max = self.device_config()

    def device_config(self) -> tuple:
        have_gpu = torch.cuda.is_available()
        if have_gpu:
            index = int(str(self.device).rsplit(":", 1)[-1]) if ":" in str(self.device) else 0
            total = torch.cuda.get_device_properties(index).total_memory
            self.gpu_mem = int(total / 1024 / 1024 / 1024 + 0.4)
        else:
            print("No supported N-card found, use CPU for inference")
            self.device, self.is_half = "cpu", True
        self.n_cpu = self.n_cpu or multiprocessing.cpu_count()
        preset = _PRESET_HALF if self.is_half else _PRESET_FULL
        if self.gpu_mem is not None and self.gpu_mem <= 4:
            preset = _PRESET_SMALL_CARD
        return preset


def _placement(device):
    """src/main.py:195 names 'cuda:0' outright and hands it to load_hubert / get_vc.  Under the host emulator of tests/ (the only
    way this package runs without a GPU) every tensor lives in host memory."""
    from . import _lib
    return torch.device("cpu") if _lib.backend() == "emu" else device


def load_hubert(device, is_half, model_path):
    """-> HuBERT content encoder in eval mode on `device` (src/rvc.py:98-109)."""
    device = _placement(device)
    net = _hubert.HubertModel(_hubert.load_state(model_path)).to(device)
    net = net.half() if is_half else net.float()
    net.eval()
    return net


def get_vc(device, is_half, config, model_path):
    """-> (checkpoint dict, version, synthesizer, target sample rate, VC pipeline object) (src/rvc.py:112-139)."""
    device = _placement(device)
    cpt = torch.load(model_path, map_location="cpu")
    if not ("config" in cpt and "weight" in cpt):
        raise ValueError(f"Incorrect format for {model_path}. Use a voice model trained using RVC v2 instead.")
    hp, params = cpt["config"], cpt["weight"]
    hp[-3] = params["emb_g.weight"].shape[0]            # number of speakers actually present in the embedding table
    tgt_sr = hp[-1]
    version = cpt.get("version", "v1")
    with_f0 = cpt.get("f0", 1) == 1
    if (version, with_f0) not in _SYNTHESIZERS:
        raise ValueError(f"{model_path}: unknown voice model version {version!r}")
    cls = _SYNTHESIZERS[(version, with_f0)]
    net_g = cls(*hp, is_half=is_half) if with_f0 else cls(*hp)
    del net_g.enc_q                                       # the posterior encoder is training-only
    print(net_g.load_state_dict(params, strict=False))
    net_g.eval().to(device)
    net_g = net_g.half() if is_half else net_g.float()
    return cpt, version, net_g, tgt_sr, VC(tgt_sr, config)


def rvc_infer(index_path, index_rate, input_path, output_path, pitch_change, f0_method, cpt, version, net_g, filter_radius,
              tgt_sr, rms_mix_rate, protect, crepe_hop_length, vc, hubert_model):
    """16 kHz mono load -> VC.pipeline -> PCM-16 WAV at tgt_sr (src/rvc.py:142-151)."""
    timings = [0, 0, 0]
    converted = vc.pipeline(hubert_model, net_g, 0, load_audio(input_path, 16000), input_path, timings, pitch_change, f0_method,
                            index_path, index_rate, cpt.get("f0", 1), filter_radius, tgt_sr, 0, rms_mix_rate, version, protect,
                            crepe_hop_length)
    wavfile.write(output_path, tgt_sr, converted)
