"""MDX-Net separation path: MDXModel.stft/istft layouts, MDX.segment/pad_wave bookkeeping (bit-exact), the U-Net on
the HIP kernels and run_mdx's arithmetic vs the oracle restatement of reference src/mdx.py."""
import numpy as np
import pytest
import torch

from aicovergen_amd.mdx import MDX, MDXModel, run_mdx_arrays
from aicovergen_amd.mdx_net import ConvTDFNet
from conftest import rel_rms
from oracle import mdxnet
from synthetic import weights
from synthetic.inputs import song_like


def _session(dev, cfg=weights.MDX_TINY, hop=64, seed=1234):
    sd = weights.mdx_state_dict(cfg, seed)
    model = MDXModel(dev.device, cfg["dim_f"], cfg["dim_t"], cfg["n_fft"], hop=hop)
    return sd, model, MDX(None, model, state_dict=sd)


def test_unet_matches_oracle(dev):
    cfg = weights.MDX_TINY
    sd = weights.mdx_state_dict(cfg, 1234)
    net = ConvTDFNet(sd, dev.device)
    for k in ("g", "n", "l", "k", "bn", "dim_f"):
        assert net.cfg[k] == cfg[k]
    torch.manual_seed(0)
    spec = torch.randn(2, 4, cfg["dim_f"], cfg["dim_t"])
    with torch.no_grad():
        ref = mdxnet.unet(sd, cfg, spec)
    assert rel_rms(net(spec), ref) < 1e-4


def test_unet_on_the_winograd_layers_matches_oracle(dev, monkeypatch):
    """The same network with every eligible 3 x 3 TFC layer on the Winograd F(2, 3) kernel (csrc/conv_ws3w.h) -- at this size the
    map-size gate would keep them on the direct kernels: the selection in ops.conv, the transformed weight image of PackedConv and
    the bias / ReLU epilogue inside the whole forward, same tolerance."""
    from aicovergen_amd import ops
    monkeypatch.setattr(ops, "winograd_min_positions", 1)
    cfg = weights.MDX_TINY
    sd = weights.mdx_state_dict(cfg, 1234)
    net = ConvTDFNet(sd, dev.device)
    assert any(pc.w_wino is not None for blk in net.ds_dense + [net.mid] + net.us_dense for pc in blk.convs)
    torch.manual_seed(0)
    spec = torch.randn(2, 4, cfg["dim_f"], cfg["dim_t"])
    with torch.no_grad():
        ref = mdxnet.unet(sd, cfg, spec)
    got = net(spec)
    monkeypatch.setattr(ops, "winograd_min_positions", 1 << 60)
    assert not torch.equal(got.cpu(), net(spec).cpu())      # the other kernels really ran
    assert rel_rms(got, ref) < 1e-4


def test_mdxmodel_stft_istft_layouts(dev):
    cfg = weights.MDX_TINY
    _, model, _ = _session(dev)
    torch.manual_seed(1)
    x = torch.randn(3, 2, model.chunk_size)
    ref = mdxnet.stft(x, cfg["n_fft"], 64, cfg["dim_f"])
    assert model.stft(x).shape == ref.shape == (3, 4, cfg["dim_f"], cfg["dim_t"])
    assert rel_rms(model.stft(x), ref) < 1e-5
    assert rel_rms(model.istft(ref), mdxnet.istft(ref, cfg["n_fft"], 64)) < 1e-5
    assert model.n_bins == cfg["n_fft"] // 2 + 1 and model.dim_c == 4
    assert tuple(model.freq_pad.shape) == (1, 4, model.n_bins - cfg["dim_f"], cfg["dim_t"])


def test_segment_and_pad_wave_bookkeeping_bit_exact(dev):
    """segment(combine=False) -> segment(combine=True) is the identity; pad_wave window count and contents equal
    the reference's (mdx.py:92-171); a length that is a multiple of gen_size gets a full extra window."""
    _, model, sess = _session(dev)
    rng = np.random.default_rng(0)
    trim = model.n_fft // 2
    gen = model.chunk_size - 2 * trim
    for n in (7000, 3 * gen, gen - 1, 1234):
        wave = rng.standard_normal((2, n)).astype(np.float32)
        for thr in (1, 2):
            segs = MDX.segment(wave, False, n // thr, 300)
            osegs = mdxnet.segment(wave, False, n // thr, 300)
            assert len(segs) == len(osegs) and all(np.array_equal(a, b) for a, b in zip(segs, osegs))
            assert np.array_equal(MDX.segment(segs, True, n // thr, 300), wave)
        mix, pad, tr = sess.pad_wave(wave)
        omix, opad, otr = mdxnet.pad_wave(wave, model.n_fft, model.chunk_size)
        assert (pad, tr) == (opad, otr) and mix.shape == omix.shape == ((n + pad) // gen, 2, model.chunk_size)
        assert torch.equal(mix.cpu(), omix)
    assert sess.pad_wave(np.zeros((2, 3 * gen), np.float32))[1] == gen


def test_process_wave_and_run_mdx_arithmetic(dev):
    """process_wave (reference semantics, 2 overlapping halves) and the denoise combination
    0.5 * (-f(-x) + f(x)) (mdx.py:261-263) vs the oracle; also the inverted stem of mdx.py:280."""
    cfg = weights.MDX_TINY
    sd, model, sess = _session(dev)
    wave = song_like(0.2, 44100, seed=3)[:, :7000]
    wave = wave / np.abs(wave).max()
    ref = mdxnet.process_wave(sd, cfg, wave, 2, hop=64)
    got = sess.process_wave(wave, 2)
    assert got.shape == ref.shape == wave.shape
    assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    refd = 0.5 * (-mdxnet.process_wave(sd, cfg, -wave, 2, hop=64) + ref)
    gotd = run_mdx_arrays(sess, wave, True, 2)
    assert np.abs(gotd - refd).max() < 1e-4 * max(1.0, np.abs(refd).max())
    assert np.abs(run_mdx_arrays(sess, wave, False, 2) - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_denoise_is_odd_part(dev):
    """Size-independent property: the denoise output is the odd part of f, so separate(-x) == -separate(x)."""
    _, _, sess = _session(dev)
    wave = song_like(0.1, 44100, seed=5)[:, :3000]
    a = run_mdx_arrays(sess, wave, True, 2)
    b = run_mdx_arrays(sess, -wave, True, 2)
    assert np.abs(a + b).max() < 1e-5 * max(1.0, np.abs(a).max())


def test_get_hash_tail(tmp_path):
    import hashlib
    p = tmp_path / "m.bin"
    blob = np.random.default_rng(0).integers(0, 255, 10000 * 1024 + 777, dtype=np.uint8).tobytes()
    p.write_bytes(blob)
    assert MDX.get_hash(str(p)) == hashlib.md5(blob[-10000 * 1024:]).hexdigest()
    p.write_bytes(blob[:1000])
    assert MDX.get_hash(str(p)) == hashlib.md5(blob[:1000]).hexdigest()


@pytest.mark.gpu
def test_voc_ft_sized_window_matches_oracle():
    """One full-size window (dim_f 3072, dim_t 256, n_fft 7680: the Voc_FT-class entry of model_data.json) through
    stft -> U-Net (g=48, 5 levels) -> istft vs the oracle on the host CPU."""
    import conftest
    conftest._bind("hip")
    cfg = weights.MDX_VOC_FT
    sd = weights.mdx_state_dict(cfg, 1234)
    model = MDXModel("cuda:0", cfg["dim_f"], cfg["dim_t"], cfg["n_fft"])
    sess = MDX(None, model, state_dict=sd)
    x = torch.from_numpy(song_like(model.chunk_size / 44100.0 + 0.01, 44100, seed=2)[:, :model.chunk_size]).unsqueeze(0)
    with torch.no_grad():
        ref = mdxnet.istft(mdxnet.unet(sd, cfg, mdxnet.stft(x, cfg["n_fft"], 1024, cfg["dim_f"])), cfg["n_fft"], 1024)
        got = model.istft_tf(sess.net.forward_tf(model.stft_tf(x.cuda())))
    assert rel_rms(got, ref) < 1e-4


def test_separation_matches_reference_golden(dev):
    """The whole separator (real hop 1024, n_fft 2048, denoise on and off) on the HIP kernels against arrays written by the
    REFERENCE's run_mdx (tests/golden/mdx_ref_tiny.npz; the network inside the reference run was the restated U-Net with the
    same seeded parameters)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mdx_ref_tiny.npz"))
    cfg = dict(weights.MDX_TINY, n_fft=2048, dim_t=16)
    _, model, sess = _session(dev, cfg, hop=1024, seed=7)
    wave = g["wave"].astype(np.float32)
    peak = max(np.max(wave), abs(np.min(wave)))
    norm = wave / peak
    scale = max(1.0, float(np.abs(g["main_plain"]).max()))
    for denoise, tag in ((True, "dn"), (False, "plain")):
        got = run_mdx_arrays(sess, norm.copy(), denoise, 2) * peak
        assert got.shape == g["main_" + tag].shape
        assert np.abs(got - g["main_" + tag]).max() < 1e-4 * scale
        inv = (-got * 1.021) + norm                      # mdx.py:280 adds the normalised wave
        assert np.abs(inv - g["inv_" + tag]).max() < 1e-4 * scale
    mix, pad, trim = sess.pad_wave(g["pad_in"])
    assert [pad, trim] == g["pad_trim"].tolist() and np.array_equal(mix.cpu().numpy(), g["pad_windows"])


@pytest.mark.parametrize("f,h,t,c,b", [(256, 64, 32, 3, 2), (768, 96, 64, 2, 1), (3072, 384, 32, 5, 1), (512, 128, 32, 2, 3)])
def test_tdf_pair_fused_kernel(dev, f, h, t, c, b):
    """aicg_tdf_pair: x + relu(bn2(relu(bn1(x W1^T + b1)) W2^T + b2)) in one launch (the f / bn intermediate stays in the
    accumulator registers) against torch and against the two-GEMM path it replaces; R = b c t rows incl. a ragged last 128-row
    workgroup, 2 / 3 / 4 / 12 accumulator tiles per wave."""
    from aicovergen_amd import ops
    torch.manual_seed(f + h)
    x = torch.randn(b, c, t, f)
    w1, b1 = torch.randn(h, f) / f ** 0.5, torch.randn(h) * 0.1
    w2, b2 = torch.randn(f, h) / h ** 0.5, torch.randn(f) * 0.1
    s1, t1, s2, t2 = torch.rand(c) + 0.5, torch.randn(c) * 0.1, torch.rand(c) + 0.5, torch.randn(c) * 0.1
    bn = lambda v, s, sh: v * s.view(1, c, 1, 1) + sh.view(1, c, 1, 1)
    ref = x + torch.relu(bn(torch.relu(bn(x @ w1.t() + b1, s1, t1)) @ w2.t() + b2, s2, t2))
    assert ops.tdf_pair_supported(f, h, t)
    d = lambda v: dev.t(v.contiguous())
    got = ops.tdf_pair(d(x), d(ops.pack_tdf_w1(w1)), d(b1), d(s1), d(t1), d(ops.pack_tdf_w2(w2)), d(b2), d(s2), d(t2))
    assert rel_rms(got, ref) < 2e-6
    two = ops.linear_last(ops.linear_last(d(x), d(w1), d(b1), d(s1), d(t1), act=ops.ACT_RELU), d(w2), d(b2), d(s2), d(t2),
                          act=ops.ACT_RELU, res=d(x))
    assert rel_rms(got, two) < 2e-6
    assert not ops.tdf_pair_supported(96, 12, 8) and not ops.tdf_pair_supported(256, 64, 48)
