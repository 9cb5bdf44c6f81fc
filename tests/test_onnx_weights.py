"""ONNX initializer reader for the MDX-Net graphs (aicovergen_amd/onnx_weights.py): protobuf decoding, the graph walk
onto kuielab ConvTDFNet parameter names, and the separator running from an `.onnx` path like the reference's
`MDX(model_path, ...)` does (src/mdx.py:74-77)."""
import os
import warnings

import numpy as np
import pytest
import torch

from aicovergen_amd import onnx_weights
from aicovergen_amd.mdx_net import ConvTDFNet, infer_cfg
from conftest import rel_rms
from oracle import mdxnet, weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIXTURE = os.path.join(GOLD, "mdx_tiny.onnx")
CFG = dict(weights.MDX_TINY)


def test_parse_committed_fixture():
    nodes, inits = onnx_weights.parse_model(FIXTURE)
    ops = [n["op"] for n in nodes]
    assert ops.count("Conv") == 14 and ops.count("ConvTranspose") == 2 and ops.count("MatMul") == 10
    assert ops.count("BatchNormalization") == 12  # tdf linears + transposed convs; Conv+BN pairs were fused by the exporter
    first = nodes[0]
    assert first["op"] == "Conv" and first["attr"]["kernel_shape"] == [1, 1]
    assert inits[first["input"][1]].shape == (CFG["g"], CFG["dim_c"], 1, 1)


def test_state_dict_reproduces_the_network():
    """Parameters read back from the graph drive the restated U-Net to the same output as the parameters that were
    exported (Conv+BN fusion by the exporter only re-associates the arithmetic)."""
    sd0 = weights.mdx_state_dict(CFG, 7)
    sd1 = onnx_weights.load_onnx_state_dict(FIXTURE)
    assert set(sd1) == {k for k in sd0 if not k.endswith("num_batches_tracked")}
    cfg1 = infer_cfg(sd1)
    assert all(cfg1[k] == CFG[k] for k in ("dim_c", "g", "n", "l", "k", "bn", "dim_f"))
    torch.manual_seed(0)
    spec = torch.randn(2, CFG["dim_c"], CFG["dim_f"], CFG["dim_t"])
    with torch.no_grad():
        assert rel_rms(mdxnet.unet(sd1, CFG, spec), mdxnet.unet(sd0, CFG, spec)) < 2e-6


def test_fresh_export_round_trip(tmp_path):
    """Same, for a graph exported now with other parameters and a different depth (n = 1, l = 3, no tdf bias)."""
    import sys
    sys.path.insert(0, GOLD)
    try:
        from make_onnx_fixture import export_unet
        cfg = dict(CFG, n=1, l=3)
        sd0 = weights.mdx_state_dict(cfg, 11)
        for k in list(sd0):
            if ".tdf." in k and k.endswith((".0.bias", ".3.bias")):
                sd0[k] = torch.zeros_like(sd0[k])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            export_unet(sd0, cfg, str(tmp_path / "m.onnx"))
    except Exception as exc:  # exporter not usable on this box: the committed fixture still covers the reader
        pytest.skip("torch.onnx.export unavailable: %r" % (exc,))
    sd1 = onnx_weights.load_onnx_state_dict(str(tmp_path / "m.onnx"))
    spec = torch.randn(1, cfg["dim_c"], cfg["dim_f"], cfg["dim_t"])
    with torch.no_grad():
        assert rel_rms(mdxnet.unet(sd1, cfg, spec), mdxnet.unet(sd0, cfg, spec)) < 2e-6


def test_kernels_run_from_onnx_path(dev):
    """ConvTDFNet built from the `.onnx` file == ConvTDFNet built from the exported state_dict, on the HIP kernels."""
    from aicovergen_amd.mdx import load_network_state
    net1 = ConvTDFNet(load_network_state(FIXTURE), dev.device)
    net0 = ConvTDFNet(weights.mdx_state_dict(CFG, 7), dev.device)
    torch.manual_seed(1)
    x = torch.randn(2, CFG["dim_c"], CFG["dim_t"], CFG["dim_f"])
    assert rel_rms(net1.forward_tf(dev.t(x)), net0.forward_tf(dev.t(x)).cpu()) < 5e-6


def test_not_a_unet_is_a_clear_error(tmp_path):
    p = tmp_path / "bad.onnx"
    p.write_bytes(b"\x08\x07")  # ModelProto with ir_version only
    with pytest.raises(ValueError, match="GraphProto"):
        onnx_weights.load_onnx_state_dict(str(p))


@pytest.mark.gpu
def test_run_mdx_end_to_end_from_onnx_and_wav_files(tmp_path):
    """The reference's own entry point, file in -> files out (src/mdx.py:238-287): model hash -> model_data entry, `.onnx`
    weights, WAV input, denoise on; main and inverted stems against the oracle's run_mdx arithmetic after PCM-16 rounding."""
    import conftest
    conftest._bind("hip")
    from aicovergen_amd import audio_io
    from aicovergen_amd.mdx import MDX, run_mdx
    from synthetic.inputs import song_like
    cfg = dict(CFG, n_fft=2048, dim_t=16)  # hop is 1024 in MDXModel: chunk = 15 360 samples
    wave = song_like(1.5, 44100, seed=9).astype(np.float32) * 0.6
    wav = tmp_path / "song.wav"
    audio_io.write_wav_pcm16(str(wav), wave.T, 44100)
    params = {MDX.get_hash(FIXTURE): {"mdx_dim_f_set": cfg["dim_f"], "mdx_dim_t_set": 4, "mdx_n_fft_scale_set": 2048,
                                      "primary_stem": "Vocals", "compensate": 1.021}}
    main, inv = run_mdx(params, str(tmp_path), FIXTURE, str(wav), denoise=True, keep_orig=True)
    assert os.path.basename(main) == "song_Vocals.wav" and os.path.basename(inv) == "song_Instrumental.wav"
    got_main, sr = audio_io.load_wav(main, 44100, mono=False)
    got_inv, _ = audio_io.load_wav(inv, 44100, mono=False)
    assert sr == 44100
    src, _ = audio_io.load_wav(str(wav), 44100, mono=False)   # what run_mdx actually read (PCM-16 rounded)
    ref_main, ref_inv = mdxnet.run_mdx_arrays(weights.mdx_state_dict(CFG, 7), cfg, src.astype(np.float64), True, 1.021, 2)
    lsb = 1.0 / 32768
    assert got_main.shape == ref_main.shape
    # the seeded random network is not gain-normalised (|out| up to ~3, the WAV writer clips like soundfile does):
    # tolerance = PCM rounding + 1e-4 of the unclipped signal range
    tol = 2 * lsb + 1e-4 * max(1.0, float(np.abs(ref_main).max()), float(np.abs(ref_inv).max()))
    assert np.abs(got_main - np.clip(ref_main, -1, 1)).max() < tol
    assert np.abs(got_inv - np.clip(ref_inv, -1, 1)).max() < tol


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("AICG_REAL_ONNX"), reason="set AICG_REAL_ONNX=<UVR .onnx> (and install onnxruntime) to pin the U-Net")
def test_real_uvr_onnx_matches_onnxruntime():
    """The one-command pin of row a2: a published UVR-MDX-NET .onnx run by onnxruntime (what the reference does, src/mdx.py:74-77,193)
    against the U-Net rebuilt from the same file's initializers on the HIP kernels, on one window."""
    import json
    import conftest
    ort = pytest.importorskip("onnxruntime")
    conftest._bind("hip")
    from aicovergen_amd.mdx import MDX, MDXModel
    path = os.environ["AICG_REAL_ONNX"]
    data = json.load(open(os.environ.get("AICG_MODEL_DATA", os.path.join(os.path.dirname(path), "model_data.json"))))
    mp = data[MDX.get_hash(path)]
    model = MDXModel("cuda:0", dim_f=mp["mdx_dim_f_set"], dim_t=2 ** mp["mdx_dim_t_set"], n_fft=mp["mdx_n_fft_scale_set"],
                     stem_name=mp["primary_stem"], compensation=mp["compensate"])
    sess = MDX(path, model)
    x = torch.randn(1, 2, model.chunk_size) * 0.1
    spec = model.stft(x.cuda())
    ref = ort.InferenceSession(path, providers=["CPUExecutionProvider"]).run(None, {"input": spec.cpu().numpy()})[0]
    got = sess.process(spec).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-3 * np.abs(ref).max()
