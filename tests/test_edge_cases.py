"""Degenerate and ragged shapes through the C ABI: empty batches, single frames, lengths just around the tile sizes."""
import pytest
import torch
import torch.nn.functional as F

from aicovergen_amd import ops
from conftest import rel_rms


def test_conv_empty_batch_and_tiny_lengths(dev):
    torch.manual_seed(0)
    w, b = torch.randn(8, 4, 3) * 0.3, torch.randn(8)
    pc = ops.PackedConv(w, b, padding=1, device=dev.device)
    y = ops.conv(dev.t(torch.zeros(0, 4, 10)), pc)
    assert tuple(y.shape) == (0, 8, 10)
    for t in (1, 2, 3, 31, 33, 127, 129, 257):
        x = torch.randn(2, 4, t)
        assert rel_rms(ops.conv(dev.t(x), pc, act=ops.ACT_RELU), F.relu(F.conv1d(x, w, b, padding=1))) < 1e-5, t


def test_conv_input_shorter_than_kernel_is_an_empty_output_or_error(dev):
    w = torch.randn(8, 4, 7)
    pc = ops.PackedConv(w, None, device=dev.device)   # no padding: T = 5 < k = 7
    with pytest.raises((RuntimeError, AssertionError)):
        ops.conv(dev.t(torch.randn(1, 4, 5)), pc)


@pytest.mark.parametrize("T", [1, 2, 31, 32, 33, 127, 129])
def test_attention_ragged_lengths(dev, T):
    torch.manual_seed(T)
    H, D = 2, 64
    q, k, v = torch.randn(H * D, T) * 0.3, torch.randn(H * D, T) * 0.3, torch.randn(H * D, T)
    qh, kh, vh = (z.view(H, D, T).transpose(1, 2) for z in (q, k, v))
    ref = (F.softmax(qh @ kh.transpose(1, 2), -1) @ vh).transpose(1, 2).reshape(H * D, T)
    assert rel_rms(ops.attention(dev.t(q), dev.t(k), dev.t(v), H), ref) < 1e-5


@pytest.mark.parametrize("T", [1, 31, 33])
def test_layernorm_single_and_ragged_frames(dev, T):
    torch.manual_seed(T)
    x, g, b = torch.randn(1, 192, T), torch.rand(192) + 0.5, torch.randn(192)
    ref = F.layer_norm(x.transpose(1, 2), (192,), g, b, 1e-5).transpose(1, 2)
    assert rel_rms(ops.layernorm_ct(dev.t(x), dev.t(g), dev.t(b)), ref) < 1e-5


def test_stft_shortest_window_round_trip(dev):
    """One hop more than the reflect padding needs: 2 frames."""
    n_fft, hop = 64, 16
    x = torch.randn(2, n_fft)  # center=True reflect padding needs len > n_fft / 2
    sp = ops.stft(dev.t(x), n_fft, hop)
    ref = torch.view_as_real(torch.stft(x, n_fft, hop, window=torch.hann_window(n_fft), center=True, return_complex=True))
    assert tuple(sp.shape) == (2, 2, n_fft // 2 + 1, 1 + x.shape[1] // hop)
    assert rel_rms(sp.cpu().permute(0, 2, 3, 1), ref) < 1e-5   # (n_sig, re/im, bins, frames) -> (n_sig, bins, frames, re/im)
    y = ops.istft(sp, n_fft, hop, x.shape[1])
    assert (y.cpu() - x).abs().max() < 2e-5
