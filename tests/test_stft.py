"""STFT / iSTFT kernels vs torch.stft / torch.istft (the calls MDXModel.stft/.istft make, reference
src/mdx.py:37-54).  Tolerance: relative RMS <= 1e-5 (fp32 FFT, different factorisation)."""
import pytest
import torch

from aicovergen_amd import ops
from conftest import rel_rms

CASES = [  # n_fft, hop, L, n_bins_out
    (7680, 1024, 1024 * 15, 3072),   # Voc_FT-class MDX parameters (model_data.json), short signal: plan 16 x 16 x 15
    (6144, 1024, 1024 * 12, 2048),   # plan 16 x 16 x 12
    (5120, 1024, 1024 * 7, 2048),    # plan 16 x 16 x 10 (radix 5 inside the registers)
    (4096, 1024, 1024 * 6 + 3, 2049),   # plan 16 x 16 x 8; odd L: the scalar overlap-add
    (8192, 1024, 1024 * 9, 3072),    # plan 16 x 16 x 16
    (16384, 1024, 1024 * 17, 4096),  # plan 16 x 16 x 32 (512-thread workgroups)
    (1024, 160, 16000, 513),         # RMVPE mel front end (src/rmvpe.py:343-345): plan 8 x 8 x 8, four frames per workgroup
    (1024, 160, 160 * 5 + 1, 513),   # ... with a ragged last workgroup (6 frames x 2 signals = 12 frames in groups of 4)
    (60, 16, 400, 31),               # no compile-time plan: run-time radix 4 * 3 * 5
    (2000, 250, 3000, 600),          # ... 4 * 2 * 5 * 5 * 5
]


def _ref_stft(x, n_fft, hop, nb):
    s = torch.stft(x, n_fft=n_fft, hop_length=hop, window=torch.hann_window(n_fft), center=True, return_complex=True)
    return torch.view_as_real(s).permute(0, 3, 1, 2)[:, :, :nb].contiguous()


@pytest.mark.parametrize("n_fft,hop,L,nb", CASES)
def test_stft_matches_torch(dev, n_fft, hop, L, nb):
    torch.manual_seed(1234)
    x = torch.randn(2, L)
    y = ops.stft(dev.t(x), n_fft, hop, nb)
    ref = _ref_stft(x, n_fft, hop, nb)
    assert y.shape == ref.shape
    assert rel_rms(y, ref) < 1e-5
    y2 = ops.stft(dev.t(x), n_fft, hop, nb, frame_major=True)
    assert torch.equal(y2.transpose(2, 3).cpu(), y.cpu())


@pytest.mark.parametrize("n_fft,hop,L,nb", CASES)
def test_istft_matches_torch(dev, n_fft, hop, L, nb):
    torch.manual_seed(4321)
    x = torch.randn(2, L)
    sp = _ref_stft(x, n_fft, hop, nb)
    sp[:, 1, 0, :] = 0.37  # a C2R transform ignores Im(DC); the UNet output has it non-zero
    out = ops.istft(dev.t(sp), n_fft, hop, L)
    pad = torch.zeros(2, 2, n_fft // 2 + 1 - nb, sp.shape[-1])
    c = torch.view_as_complex(torch.cat([sp, pad], 2).permute(0, 2, 3, 1).contiguous())
    ref = torch.istft(c, n_fft=n_fft, hop_length=hop, window=torch.hann_window(n_fft), center=True, length=L)
    assert out.shape[1] == L and ref.shape[1] == L
    assert rel_rms(out, ref) < 1e-5


def test_stft_istft_roundtrip_full_band(dev):
    """Size-independent property: full-band STFT -> iSTFT is the identity (reference: 9.5e-7)."""
    torch.manual_seed(7)
    n_fft, hop = 7680, 1024
    L = hop * (255 if dev.big else 20)
    x = torch.randn(4 if dev.big else 2, L)
    sp = ops.stft(dev.t(x), n_fft, hop)
    y = ops.istft(sp, n_fft, hop, L)
    assert (y.cpu() - x).abs().max() < 2e-5


def test_stft_rejects_bad_shapes(dev):
    x = dev.t(torch.zeros(1, 100))
    with pytest.raises(RuntimeError):
        ops.stft(x, 14, 4)  # 7 is not 2^a 3^b 5^c
    with pytest.raises(RuntimeError):
        ops.stft(dev.t(torch.zeros(1, 10)), 64, 16)  # reflect padding needs L > n_fft/2
