"""Writes tests/golden/mdx_ref_tiny.npz by running the REFERENCE's own src/mdx.py (MDXModel.stft / istft, MDX.segment /
pad_wave / _process_wave / process_wave and run_mdx) in the build container.

The reference needs onnxruntime, librosa and soundfile, none of which is installed here, and a CUDA device.  Only those
edges are replaced, by stubs defined in this script:
  * onnxruntime.InferenceSession -> runs the restated TFC-TDF U-Net (oracle/mdxnet.py: unet) with the seeded MDX_TINY
    parameters; everything AROUND the network (framing, STFT, threads, denoise, inversion) is the reference's code;
  * librosa.load -> returns the synthetic stereo wave; soundfile.write -> captures the arrays run_mdx writes;
  * MDX.__init__'s default processor is switched from 0 to -1 (CPU device / CPUExecutionProvider) and
    torch.cuda.get_device_properties is faked as a 16 GB card (run_mdx queries it to pick m_threads = 2).
Run from the repo root:  python tests/golden/make_mdx_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import mdxnet  # noqa: E402
from synthetic import weights
from synthetic.inputs import song_like  # noqa: E402

CFG = dict(weights.MDX_TINY, n_fft=2048, dim_t=16)
SD = weights.mdx_state_dict(weights.MDX_TINY, 7)
captured = {}


def install_stubs(wave):
    ort = types.ModuleType("onnxruntime")

    class InferenceSession:
        def __init__(self, path, providers=None):
            self.providers = providers

        def run(self, _names, feed):
            with torch.no_grad():
                return [mdxnet.unet(SD, CFG, torch.from_numpy(np.asarray(feed["input"], dtype=np.float32))).numpy()]

    ort.InferenceSession = InferenceSession
    sys.modules["onnxruntime"] = ort
    lib = types.ModuleType("librosa")
    lib.load = lambda filename, mono=False, sr=44100: (wave.copy(), sr)
    sys.modules["librosa"] = lib
    sf = types.ModuleType("soundfile")

    def write(path, data, sr):
        captured[os.path.basename(path)] = np.array(data, dtype=np.float64)

    sf.write = write
    sys.modules["soundfile"] = sf
    torch.cuda.get_device_properties = lambda device: types.SimpleNamespace(total_memory=16 * 1024 ** 3)


def main():
    wave = (song_like(0.8, 44100, seed=9) * 0.6).astype(np.float32)      # (2, 35280), peak < 1
    install_stubs(wave)
    sys.path.insert(0, "/root/reference/src")
    import mdx as ref                                                      # the reference module itself
    ref.MDX.__init__.__defaults__ = (-1,)                                  # processor: CPU
    model_file = os.path.join(HERE, "mdx_tiny.onnx")                       # only hashed by run_mdx (the stub ignores its content)
    params = {ref.MDX.get_hash(model_file): {"mdx_dim_f_set": CFG["dim_f"], "mdx_dim_t_set": 4, "mdx_n_fft_scale_set": 2048,
                                             "primary_stem": "Vocals", "compensate": 1.021}}
    out = {"wave": wave}
    for denoise in (True, False):
        captured.clear()
        main_path, inv_path = ref.run_mdx(params, "/tmp", model_file, "/tmp/song.wav", denoise=denoise, keep_orig=True)
        assert os.path.basename(main_path) == "song_Vocals.wav" and os.path.basename(inv_path) == "song_Instrumental.wav"
        tag = "dn" if denoise else "plain"
        out["main_" + tag] = captured["song_Vocals.wav"].T.astype(np.float32)          # (2, N)
        out["inv_" + tag] = captured["song_Instrumental.wav"].T.astype(np.float32)
    # pieces: model-level STFT / iSTFT and the window bookkeeping on a ragged length
    model = ref.MDXModel(torch.device("cpu"), dim_f=CFG["dim_f"], dim_t=CFG["dim_t"], n_fft=CFG["n_fft"])
    sess = ref.MDX(model_file, model)
    x = torch.from_numpy(wave[:, :model.chunk_size].copy())[None]
    spec = model.stft(x)
    out["stft_in"], out["stft_out"] = x.numpy(), spec.numpy()
    out["istft_out"] = model.istft(spec).numpy()
    ragged = wave[:, :30001].astype(np.float64)
    mix, pad, trim = sess.pad_wave(ragged)
    out["pad_in"], out["pad_windows"], out["pad_trim"] = ragged, mix.numpy(), np.array([pad, trim])
    segs = sess.segment(ragged, False, 15000)
    out["seg_lens"] = np.array([s.shape[-1] for s in segs])
    out["seg_joined"] = sess.segment(segs, True, 15000)
    out["process_wave"] = sess.process_wave(wave / np.abs(wave).max(), 2).astype(np.float32)
    path = os.path.join(HERE, "mdx_ref_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
