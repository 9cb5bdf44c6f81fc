"""Thin torch-tensor wrappers over the C ABI (include/aicg.h).  No arithmetic happens here: these
functions validate shapes, allocate outputs with torch, and hand raw device pointers + the current HIP
stream to libaicg_hip.so."""
import contextlib
import functools
import math
import os
import threading

import numpy as np
import torch
from . import _env

from . import _lib


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream(t):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return 0


_tls = threading.local()


def _check(*tensors):
    """All tensors must be fp32/int, contiguous where required by the caller, and live where the bound
    library executes: HIP device memory of ONE device for the product library, host memory for the test emulator.
    Remembers that device for the launch that follows (`_call`)."""
    be = _lib.backend()
    dev = None
    for t in tensors:
        if t is None:
            continue
        if be == "hip" and not t.is_cuda:
            raise RuntimeError("aicovergen_amd: tensor on %s, but the HIP kernels need device memory" % t.device)
        if be == "emu" and t.is_cuda:
            raise RuntimeError("emulator backend needs host tensors")
        if t.is_cuda:
            if dev is None:
                dev = t.device.index
            elif dev != t.device.index:
                raise RuntimeError("aicovergen_amd: operands on different devices (cuda:%d and cuda:%d)" % (dev, t.device.index))
    _tls.dev = dev


def _call(name, *args):
    """_lib.call with the operands' device made current: the C ABI launches on the calling thread's current HIP device, and
    the stream / pointers handed over belong to the tensors' device (MDX(processor=1), Config('cuda:1') never call
    torch.cuda.set_device)."""
    dev = getattr(_tls, "dev", None)
    if dev is not None and dev != torch.cuda.current_device():
        with torch.cuda.device(dev):
            return _lib.call(name, *args)
    return _lib.call(name, *args)


# ---------------------------------------------------------------------------------------------------
# STFT / iSTFT
# ---------------------------------------------------------------------------------------------------
_fft_tables = {}


def fft_tables(n_fft, device, window=None):
    """(window, tw_half, tw_full) on `device`; twiddles are evaluated in float64 and rounded once."""
    key = (n_fft, str(device))
    if key not in _fft_tables:
        m = n_fft // 2
        k = np.arange(m, dtype=np.float64)
        th = np.exp(-2j * np.pi * k / m)
        k2 = np.arange(m + 1, dtype=np.float64)
        tf = np.exp(-2j * np.pi * k2 / n_fft)
        tw_half = torch.from_numpy(np.stack([th.real, th.imag], -1).astype(np.float32)).contiguous().to(device)
        tw_full = torch.from_numpy(np.stack([tf.real, tf.imag], -1).astype(np.float32)).contiguous().to(device)
        hann = torch.hann_window(n_fft, periodic=True, dtype=torch.float32).to(device)
        _fft_tables[key] = (hann, tw_half, tw_full)
    hann, tw_half, tw_full = _fft_tables[key]
    if window is None:
        window = hann
    return window, tw_half, tw_full


def stft(x, n_fft, hop, n_bins_out=None, frame_major=False, window=None):
    """x: (n_sig, L) fp32.  Returns the complex spectrogram split into planes:
    frame_major=False -> (n_sig, 2, n_bins_out, n_frames)  [torch.stft + view_as_real + permute layout]
    frame_major=True  -> (n_sig, 2, n_frames, n_bins_out)  [bins contiguous: coalesced stores]"""
    assert x.dim() == 2 and x.dtype == torch.float32
    x = x.contiguous()
    n_sig, L = x.shape
    n_frames = 1 + L // hop
    nb = n_fft // 2 + 1 if n_bins_out is None else n_bins_out
    window, tw_half, tw_full = fft_tables(n_fft, x.device, window)
    _check(x, window)
    if frame_major:
        out = torch.empty((n_sig, 2, n_frames, nb), dtype=torch.float32, device=x.device)
        o_bin, o_frame = 1, nb
    else:
        out = torch.empty((n_sig, 2, nb, n_frames), dtype=torch.float32, device=x.device)
        o_bin, o_frame = n_frames, 1
    _call("aicg_stft", _ptr(x), _ptr(out), _ptr(window), _ptr(tw_half), _ptr(tw_full), n_sig, L, n_fft, hop,
              n_frames, nb, 2 * nb * n_frames, nb * n_frames, o_bin, o_frame, _stream(x))
    return out


def istft(spec, n_fft, hop, length, frame_major=False, window=None):
    """spec: (n_sig, 2, n_bins_in, n_frames) (or (n_sig, 2, n_frames, n_bins_in) if frame_major);
    bins above n_bins_in are zero.  Returns (n_sig, length)."""
    assert spec.dim() == 4 and spec.shape[1] == 2 and spec.dtype == torch.float32
    spec = spec.contiguous()
    n_sig = spec.shape[0]
    if frame_major:
        n_frames, nb = spec.shape[2], spec.shape[3]
        i_bin, i_frame = 1, nb
    else:
        nb, n_frames = spec.shape[2], spec.shape[3]
        i_bin, i_frame = n_frames, 1
    window, tw_half, tw_full = fft_tables(n_fft, spec.device, window)
    _check(spec, window)
    frames = torch.empty((n_sig, n_frames, n_fft), dtype=torch.float32, device=spec.device)
    out = torch.empty((n_sig, length), dtype=torch.float32, device=spec.device)
    st = _stream(spec)
    _call("aicg_istft_frames", _ptr(spec), _ptr(frames), _ptr(window), _ptr(tw_half), _ptr(tw_full), n_sig,
              n_fft, n_frames, nb, 2 * nb * n_frames, nb * n_frames, i_bin, i_frame, st)
    _call("aicg_istft_ola", _ptr(frames), _ptr(window), _ptr(out), n_sig, length, n_fft, hop, n_frames, st)
    return out


# ---------------------------------------------------------------------------------------------------
# Convolution (implicit GEMM on fp32 MFMA)
# ---------------------------------------------------------------------------------------------------
import ctypes

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4, 5


class ConvDesc(ctypes.Structure):
    """Mirror of aicg_conv_desc (include/aicg.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "Cin", "H", "W", "Cout", "Ho", "Wo", "KH", "KW", "stride_h", "stride_w", "pad_h", "pad_w",
                 "dil_h", "dil_w", "groups")] + \
               [(n, ctypes.c_int64) for n in ("x_sn", "x_sc", "x_sh", "y_sn", "y_sc", "y_sh", "r_sn", "r_sc", "r_sh")] + \
               [("pre_act", ctypes.c_int32), ("pre_slope", ctypes.c_float), ("act", ctypes.c_int32),
                ("act_slope", ctypes.c_float), ("out_scale", ctypes.c_float), ("accumulate", ctypes.c_int32),
                ("res_before_act", ctypes.c_int32), ("pad_h_end", ctypes.c_int32), ("pad_w_end", ctypes.c_int32),
                ("shuffle", ctypes.c_int32), ("res_mul", ctypes.c_int32), ("packed_v3", ctypes.c_int32),
                ("split", ctypes.c_int32), ("wino", ctypes.c_int32), ("gemm_tile", ctypes.c_int32)]


def conv_bkc(taps):
    return _lib.get().aicg_conv_bkc(int(taps))


# Opt-in split precision for the convolution family (AICG_PRECISION=bf16x3; default fp32 on the fp32 MFMA): layers packed while
# this is set carry a third weight image and run csrc/conv_ws3s.h.  Read once at import; tests flip the module attribute.
split_precision = os.environ.get("AICG_PRECISION", "fp32").lower() in ("bf16x3", "split")


# Opt-in fp16 matrix arithmetic for the RVC half (AICG_HALF=1 AND the caller's is_half=True, i.e. what src/main.py:196 asks for and the
# reference runs in fp16 on a GPU, src/rvc.py:103-104,137-138): HubertModel.half() / Synthesizer.half() mark their layers (mark_half), and a
# marked layer that takes one of the LDS-DMA staged kernels (csrc/conv_g1.h, csrc/conv_g1w.h) rounds its operands to fp16 in registers in
# front of the MFMA -- fp32 activations in HBM, fp32 accumulation, fp32 epilogue.  The f0 models, SineGen and every layer on another
# kernel stay fp32.  Default (and the headline bench): off -- .half() is then the no-op it was.
def half_requested():
    return os.environ.get("AICG_HALF", "0") == "1"


def mark_half(tree, on=True):
    """Set / clear the fp16-operand flag on every PackedConv reachable from `tree` (dicts, lists, tuples, PackedConvTranspose)."""
    if isinstance(tree, PackedConv):
        tree.f16 = bool(on) and not tree.fp32_only and not tree.split
    elif isinstance(tree, PackedConvTranspose):
        mark_half(tree.gemm, on)
    elif isinstance(tree, dict):
        for v in tree.values():
            mark_half(v, on)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            mark_half(v, on)


_fp32_depth = 0


@contextlib.contextmanager
def fp32_layers():
    """Layers packed inside this block stay on the fp32 MFMA whatever AICG_PRECISION says: the f0 estimators and the retrieval
    search select INDICES (pitch bins, neighbour ids), which are kept bit-exact (BASELINE north_star)."""
    global split_precision, _fp32_depth
    old, split_precision = split_precision, False
    _fp32_depth += 1
    try:
        yield
    finally:
        split_precision = old
        _fp32_depth -= 1


def fp32_only(fn):
    """Decorator form of fp32_layers for constructors."""
    @functools.wraps(fn)
    def wrapped(*a, **k):
        with fp32_layers():
            return fn(*a, **k)
    return wrapped


def _split_image(out):
    """(groups, taps, Cin_pad, Mpad) fp32 -> the bf16 hi / lo image [tap][Cin_pad/16][hi|lo][h][Mpad][8], as fp32 words."""
    g, taps, cpad, mpad = out.shape
    hi = out.to(torch.bfloat16)                       # round to nearest even
    lo = (out - hi.to(torch.float32)).to(torch.bfloat16)
    both = torch.stack([hi, lo], 0)                   # (part, g, taps, cpad, mpad)
    both = both.reshape(2, g, taps, cpad // 16, 2, 8, mpad).permute(1, 2, 3, 0, 4, 6, 5).contiguous()
    return both.view(torch.int16).reshape(-1).view(torch.float32)


def pack_conv_weight(w, groups=1, split=False):
    """(Cout, Cin/groups, KH, KW) -> two images back to back (pure re-layout + zero pad; Cin_pad and Mpad are multiples of
    32): per group [tap][Cin_pad][Mpad], then per group [tap][Cin_pad/8][2][Mpad][4] (aicg_conv_desc.packed_v3); with `split` a
    third one of the same size, the bf16 hi / lo pairs of aicg_conv_desc.split."""
    w = w.detach().to(torch.float32)
    cout, cin_g, kh, kw = w.shape
    taps = kh * kw
    cpad = (cin_g + 31) // 32 * 32
    cout_g = cout // groups
    mpad = (cout_g + 31) // 32 * 32
    out = torch.zeros((groups, taps, cpad, mpad), dtype=torch.float32, device=w.device)
    out[:, :, :cin_g, :cout_g] = w.reshape(groups, cout_g, cin_g, taps).permute(0, 3, 2, 1)
    # second image for the 16-byte-fragment kernels (csrc/conv_ws3.h): input channel ci = 8 q + 2 j + parity is element j of the
    # quad at [tap][q][parity][m]
    v3 = out.reshape(groups, taps, cpad // 8, 4, 2, mpad).permute(0, 1, 2, 4, 5, 3)
    parts = [out.reshape(-1), v3.reshape(-1)]
    if split:
        parts.append(_split_image(out))
    return torch.cat(parts).contiguous()


# Winograd F(2, 3) along rows for plain 3 x 3 layers (csrc/conv_ws3w.h): 12 instead of 18 MFMA contractions per output pair.  Layers
# packed while this is set carry the transformed kernel next to the direct one; conv() takes it for large maps with a bias + none /
# ReLU epilogue.  The f0 models (packed under fp32_layers()) never do: their goldens pin the direct summation order's bins.
winograd = _env.dev("AICG_WINOGRAD", "1") != "0"
winograd_min_positions = 32768   # below this the 64-column tiles (4 or 8 rows) do not fill the chip (tests lower it)


def winograd_kernel(w):
    """(Cout, Cin, 3, 3) -> (Cout, Cin, 3, 4): G g along the last axis, G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]]."""
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]
    return torch.stack([g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2], -1).contiguous()


# The two-dimensional form F(2 x 2, 3 x 3) (csrc/conv_w2d.h): 16 instead of 36 contractions per 2 x 2 output block, for layers whose
# output channels come in units of 48 (every MDX-Net level).  AICG_WINOGRAD=1 keeps the row-only form everywhere.
winograd2d = _env.dev("AICG_WINOGRAD", "2") == "2"
winograd2d_waves = int(_env.dev("AICG_W2D_WAVES", "8"))   # 8: two waves per SIMD on an 8 x 64 tile; 4: one per SIMD on 4 x 64


winograd2d_code = int(_env.dev("AICG_W2D_CODE", "0"))       # tools: a schedule variant under test (aicg_conv_desc.wino 6 ..)
winograd2d_quads = _env.dev("AICG_W2D_QUADS", "0") == "1"   # fragment image: [s][p / 4][ks][m][p % 4] (16-byte fragments)
# the eight-wave form reads PAIR fragments ([s][p / 2][ks][m][p % 2], one ds_read_b64 per two MFMAs: aicg_conv_desc.wino 12) unless
# AICG_W2D_PAIRS=0 (dword fragments, wino 2): 1-6 % faster on the five MDX-Net levels (profiles/r06_kbench_w2d_pairs.txt)
winograd2d_pairs = _env.dev("AICG_W2D_PAIRS", "1") != "0"


def _w2d_code():
    """aicg_conv_desc.wino of a layer on the two-dimensional form, and which image it reads ("dword" / "pairs" / "quads")."""
    code = winograd2d_code or ((4 if winograd2d_quads else 12 if winograd2d_pairs else 2) if winograd2d_waves == 8
                               else (5 if winograd2d_quads else 3))
    return code, ("pairs" if code in (12, 13, 14, 15, 16) else "quads" if code in (4, 5) else "dword")


def winograd2d_image(w, quads=False, pairs=False):
    """(Cout, Cin, 3, 3), Cout % 48 == 0 -> the image aicg_conv_desc.wino == 2 reads (include/aicg.h): U = G g G^T per (co, ci), laid out
    [Cout / 48][ceil(Cin / 8)][s][point 4 i + q][ks][m] with input channel 8 chunk + 4 s + ks (zero beyond Cin); `quads`:
    [Cout / 48][ceil(Cin / 8)][s][point / 4][ks][m][point % 4] (wino == 4 / 5)."""
    co, ci = w.shape[:2]
    assert co % 48 == 0 and tuple(w.shape[2:]) == (3, 3)
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w.device)
    U = torch.einsum("ia,ocab,jb->ocij", G, w.double(), G).to(torch.float32)          # exact: every factor is a power of two
    cpad = (ci + 7) // 8 * 8
    Up = torch.zeros((co, cpad, 16), dtype=torch.float32, device=w.device)
    Up[:, :ci] = U.reshape(co, ci, 16)
    if quads:
        return Up.view(co // 48, 48, cpad // 8, 2, 4, 4, 4).permute(0, 2, 3, 5, 4, 1, 6).contiguous().view(-1)
    if pairs:   # [Cout / 48][ceil(Cin / 8)][s][point / 2][ks][m][point % 2] (wino == 12)
        return Up.view(co // 48, 48, cpad // 8, 2, 4, 8, 2).permute(0, 2, 3, 5, 4, 1, 6).contiguous().view(-1)
    return Up.view(co // 48, 48, cpad // 8, 2, 4, 16).permute(0, 2, 3, 5, 4, 1).contiguous().view(-1)


# One-dimensional Winograd F(2, 3) for the vocoder's k = 3 / 7 / 11 ResBlock layers, dilation 1 / 3 / 5 (csrc/conv_g1w.h): 4 / 10 / 15 products
# per output pair and input channel instead of 6 / 14 / 22.  Layers packed while this is set carry the slot image next to the direct one;
# conv() takes it where the kernel applies (aligned rows, W % 4 == 0, enough positions).  The f0 models never do (fp32_layers()).
winograd1d = _env.dev("AICG_WINOGRAD1D", "1") != "0"
winograd1d_min_positions = int(_env.dev("AICG_WINOGRAD1D_MIN", "16384"))


def winograd1d_kernel(w):
    """(Cout, Cin, K), K in {3, 7, 11} -> (Cout, Cin, S) slot weights of csrc/conv_g1w.h, S = 4 / 10 / 15: per 3-tap group (g0, g1, g2)
    the four F(2, 3) weights (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2); a remainder of two taps (a, b): (a, a + b, b); one tap a:
    (a, -a).  Sums in float64, rounded once."""
    k = w.shape[-1]
    assert k in (3, 5, 7, 11)
    wd = w.detach().double()
    slots = []
    for g in range(k // 3):
        g0, g1, g2 = wd[..., 3 * g], wd[..., 3 * g + 1], wd[..., 3 * g + 2]
        slots += [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]
    if k % 3 == 2:
        a, b = wd[..., k - 2], wd[..., k - 1]
        slots += [a, a + b, b]
    elif k % 3 == 1:
        a = wd[..., k - 1]
        slots += [a, -a]
    return torch.stack(slots, -1).to(torch.float32).contiguous()


class PackedConv:
    """A convolution layer ready for aicg_conv_forward: packed weights + geometry.  1-D layers use KH=1."""

    def __init__(self, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, device=None, padding_end=None):
        self.padding_end = None if padding_end is None else (0, _one(padding_end)) if weight.dim() == 3 else _pair(padding_end)
        if weight.dim() == 3:  # Conv1d (Cout, Cin_g, K)
            weight = weight.unsqueeze(2)
            stride, padding, dilation = (1, _one(stride)), (0, _one(padding)), (1, _one(dilation))
        elif weight.dim() == 2:  # Linear (out, in)
            weight = weight[:, :, None, None]
            stride, padding, dilation = (1, 1), (0, 0), (1, 1)
        else:
            stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.cout, cin_g, self.kh, self.kw = weight.shape
        self.cin = cin_g * groups
        self.groups = groups
        self.stride, self.padding, self.dilation = stride, padding, dilation
        device = weight.device if device is None else device
        self.split = bool(split_precision)
        self.fp32_only = _fp32_depth > 0                    # packed inside fp32_layers(): never marked for fp16 operands
        self.f16 = False                                    # mark_half()
        self.w = pack_conv_weight(weight.to(device), groups, self.split)
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.w_wino = self.w_wino2 = self.w_wino1 = None
        if (winograd1d and not self.split and not _fp32_depth and self.kh == 1 and self.kw in (3, 5, 7, 11) and stride == (1, 1)
                and dilation[1] in ((1, 3, 5) if self.kw != 5 else (1,)) and padding == (0, (self.kw - 1) // 2 * dilation[1]) and self.padding_end is None
                and groups == 1 and cin_g >= 16):
            # slot image of the 1-D Winograd form, packed like any k-tap kernel (slots in the taps' place)
            self.w_wino1 = pack_conv_weight(winograd1d_kernel(weight.detach().to(device=device, dtype=torch.float32)[:, :, 0]).unsqueeze(2), 1, False)
        if (winograd and not self.split and not _fp32_depth and (self.kh, self.kw) == (3, 3) and stride == (1, 1) and dilation == (1, 1)
                and padding == (1, 1) and self.padding_end is None and groups == 1 and cin_g >= 8):
            self.w_wino = pack_conv_weight(winograd_kernel(weight.detach().to(device=device, dtype=torch.float32)), 1, False)
            if winograd2d and self.cout % 48 == 0:
                # ONE image on the device: the one the routed kernel reads (_w2d_code()).  The other fragment layouts (16/9 of the weights
                # in HBM each) exist only for the dev switches / A-B tools: built on first use from the caller's own weight tensor (a
                # reference, normally host memory), never for the product's layers
                self._wino2_src, self._wino2_dev, self._wino2_images = weight.detach(), device, {}
                self.w_wino2 = self.wino2_image(_w2d_code()[1])

    def wino2_image(self, kind):
        """The F(2 x 2, 3 x 3) weight image with "dword" / "pairs" / "quads" fragments (winograd2d_image)."""
        if kind not in self._wino2_images:
            self._wino2_images[kind] = winograd2d_image(self._wino2_src.to(device=self._wino2_dev, dtype=torch.float32),
                                                         quads=kind == "quads", pairs=kind == "pairs")
        return self._wino2_images[kind]

    def wino2q(self):
        return self.wino2_image("quads")

    def out_hw(self, h, w):
        pe = self.padding if self.padding_end is None else self.padding_end
        ho = (h + self.padding[0] + pe[0] - self.dilation[0] * (self.kh - 1) - 1) // self.stride[0] + 1
        wo = (w + self.padding[1] + pe[1] - self.dilation[1] * (self.kw - 1) - 1) // self.stride[1] + 1
        return ho, wo


def _one(v):
    return int(v[0]) if isinstance(v, (tuple, list)) else int(v)


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def _as4d(t):
    return t.unsqueeze(2) if t.dim() == 3 else t


class ConvProfile:
    """Optional per-launch accounting of the conv kernel (bench.py roofline): algorithmic FLOPs and HIP-event time of
    every aicg_conv_forward launch on the current stream.  Enabled with `ops.conv_profile = ConvProfile()`."""

    def __init__(self):
        self.events = []
        self.flops = 0.0
        self.flops_executed = 0.0  # what the matrix pipe was given (the Winograd forms execute fewer multiply-adds than the direct form)
        self.bytes = 0.0  # algorithmic HBM bytes: input + packed weights + output (+ residual / accumulate reads), each once
        self.launches = 0
        self.shapes = []  # per launch: (shape key, flops) -- by_shape() groups them

    def by_shape(self):
        """[{shape, launches, ms, tflops, gflop}] sorted by time: where the family's time goes (bench.py --conv-shapes)."""
        torch.cuda.synchronize()
        agg = {}
        for (key, fl), (a, b) in zip(self.shapes, self.events):
            r = agg.setdefault(key, [0, 0.0, 0.0])
            r[0] += 1
            r[1] += a.elapsed_time(b)
            r[2] += fl
        rows = [{"shape": k, "launches": v[0], "ms": v[1], "gflop": v[2] / 1e9, "tflops": v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0}
                for k, v in agg.items()]
        return sorted(rows, key=lambda r: -r["ms"])

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.events)
        return {"launches": self.launches, "flops": self.flops, "flops_executed": self.flops_executed, "ms": ms, "bytes": self.bytes,
                "tflops": (self.flops / (ms * 1e-3) / 1e12) if ms > 0 else 0.0,
                "tflops_executed": (self.flops_executed / (ms * 1e-3) / 1e12) if ms > 0 else 0.0}


conv_profile = None

# tools: aicg_conv_desc.gemm_tile of every launch (0 = the library's policy; 1 keeps 1 x 1 layers off csrc/conv_g1.h; 2 / 3 / 4 force a tile)
gemm_tile = 0


def conv(x, pc, res=None, out=None, pre_act=ACT_NONE, pre_slope=0.0, act=ACT_NONE, act_slope=0.0, out_scale=1.0,
         accumulate=False, bias=None, res_before_act=False, out_len=None, shuffle=0, res_mul=False):
    """y = [y +] out_scale * (act(conv(pre_act(x)) + bias) + res).  x: (N,C,T) or (N,C,H,W), last dim contiguous;
    views with arbitrary batch/channel/row strides are accepted for x, res and out."""
    is1d = x.dim() == 3
    x4 = _as4d(x)
    n, c, h, w = x4.shape
    assert c == pc.cin, "conv: input has %d channels, layer expects %d" % (c, pc.cin)
    assert x4.stride(3) == 1 or w == 1
    # the kernels' range-checked buffer loads bound a channel chunk by its channel stride: a channel's rows must lie inside it
    assert h == 1 or c == 1 or x4.stride(1) >= x4.stride(2) * (h - 1) + w, "conv: input rows must lie inside the channel stride"
    ho, wo = pc.out_hw(h, w)
    if out_len is not None:  # compute only the first out_len columns (e.g. SamePad of an even kernel)
        assert out_len <= wo
        wo = out_len
    # shuffle = 2: this is the GEMM of a kernel = stride = 2 ConvTranspose2d and the epilogue scatters row m of position
    # (ho, wo) to out[m >> 2][2 ho + ((m >> 1) & 1)][2 wo + (m & 1)] (see conv_transpose)
    oshape = (n, pc.cout // 4, 2 * ho, 2 * wo) if shuffle else (n, pc.cout, ho, wo)
    if out is None:
        out = torch.empty((n, pc.cout, wo) if is1d else oshape, dtype=torch.float32, device=x.device)
        assert not accumulate
    o4 = _as4d(out)
    assert o4.shape == oshape, (o4.shape, oshape)
    assert o4.stride(3) == 1 or wo == 1
    r4 = None
    if res is not None:
        r4 = _as4d(res)
        assert r4.shape == o4.shape and (r4.stride(3) == 1 or wo == 1)
    b = pc.bias if bias is None else bias
    if o4.numel() == 0:  # empty batch / no output positions: nothing to launch (empty tensors have no storage to point at)
        return out
    _check(x, out, res, pc.w, b)
    d = ConvDesc()
    d.N, d.Cin, d.H, d.W, d.Cout, d.Ho, d.Wo = n, c, h, w, pc.cout, ho, wo
    d.KH, d.KW = pc.kh, pc.kw
    d.stride_h, d.stride_w = pc.stride
    d.pad_h, d.pad_w = pc.padding
    d.dil_h, d.dil_w = pc.dilation
    d.groups = pc.groups
    d.x_sn, d.x_sc, d.x_sh = x4.stride(0), x4.stride(1), x4.stride(2)
    d.y_sn, d.y_sc, d.y_sh = o4.stride(0), o4.stride(1), o4.stride(2)
    if r4 is not None:
        d.r_sn, d.r_sc, d.r_sh = r4.stride(0), r4.stride(1), r4.stride(2)
    d.pre_act, d.pre_slope, d.act, d.act_slope = pre_act, pre_slope, act, act_slope
    d.out_scale, d.accumulate = out_scale, 1 if accumulate else 0
    d.res_before_act = 1 if res_before_act else 0
    d.pad_h_end, d.pad_w_end = (-1, -1) if pc.padding_end is None else pc.padding_end
    d.shuffle, d.res_mul = int(shuffle), 1 if res_mul else 0
    d.packed_v3 = 1
    d.split = 1 if getattr(pc, "split", False) else 2 if getattr(pc, "f16", False) else 0
    # Winograd form: plain 3 x 3 layer, bias + none / ReLU epilogue, an even row length and enough output to fill the chip
    wino = (getattr(pc, "w_wino", None) is not None and not is1d and res is None and not accumulate and pre_act == ACT_NONE
            and out_scale == 1.0 and act in (ACT_NONE, ACT_RELU) and not shuffle and out_len is None and wo % 2 == 0
            and n * ho * wo >= winograd_min_positions)
    # ... its two-dimensional form where the layer has it and the map is float4-aligned
    wino2 = (wino and winograd2d and getattr(pc, "w_wino2", None) is not None and w % 4 == 0 and x4.stride(0) % 4 == 0 and x4.stride(1) % 4 == 0
             and x4.stride(2) % 4 == 0 and x4.data_ptr() % 16 == 0)
    # the one-dimensional form F(2, 3) of a k = 3 / 7 / 11 layer (csrc/conv_g1w.h): aligned rows of a multiple of four positions
    # (layers of fewer than 16 output channels -- the vocoder's conv_post, 32 -> 1 -- are one HBM pass: the streaming kernels keep them)
    wino1 = (winograd1d and getattr(pc, "w_wino1", None) is not None and pc.cout >= 16 and is1d and not shuffle and out_len is None and w % 4 == 0
             and n * w >= winograd1d_min_positions and pre_act in (ACT_NONE, ACT_LRELU) and 0.0 <= pre_slope <= 1.0
             and not res_mul and w < (1 << 24) and x4.stride(1) < (1 << 24)      # what csrc/conv_g1w.h's conv_g1w_applicable() also demands
             and 15 * (-(-(c // pc.groups) // 32) * 32) * (-(-pc.cout // 32) * 32) * 4 < (1 << 31)
             and x4.data_ptr() % 16 == 0 and x4.stride(0) % 4 == 0 and x4.stride(1) % 4 == 0 and x4.stride(1) >= w
             and o4.data_ptr() % 16 == 0 and o4.stride(0) % 4 == 0 and o4.stride(1) % 4 == 0
             and (r4 is None or (r4.data_ptr() % 16 == 0 and r4.stride(0) % 4 == 0 and r4.stride(1) % 4 == 0)))
    # 2 / 3: eight / four waves per workgroup; + 2: quad fragments
    d.gemm_tile = gemm_tile
    w2code, w2kind = _w2d_code()
    d.wino = w2code if wino2 else 1 if wino else 8 if wino1 else 0
    prof = conv_profile
    if prof is not None and x.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _call("aicg_conv_forward", ctypes.addressof(d), _ptr(x4), _ptr(pc.wino2_image(w2kind) if wino2 else pc.w_wino if wino else pc.w_wino1 if wino1 else pc.w), _ptr(b), _ptr(r4), _ptr(o4), _stream(x))
    if prof is not None and x.is_cuda:
        e1.record()
        prof.events.append((e0, e1))
        prof.flops += 2.0 * n * pc.cout * (pc.cin // pc.groups) * pc.kh * pc.kw * ho * wo
        # what the matrix pipe is given: F(2 x 2, 3 x 3) 16 of 36 multiply-adds, the row form 12 of 18, the 1-D form 4 / 7 / 10 / 15 slots per
        # output PAIR where the direct form spends 2 k (k = 3 / 5 / 7 / 11) -- times 32 / 30 for the dilated layers' two idle lanes per tile
        exe = 4.0 / 9.0 if wino2 else 2.0 / 3.0 if wino else 1.0
        if wino1:
            exe = {3: 4, 5: 7, 7: 10, 11: 15}[pc.kw] / (2.0 * pc.kw) * (1.0 if pc.dilation[1] == 1 else 32.0 / 30.0)
        prof.flops_executed += 2.0 * n * pc.cout * (pc.cin // pc.groups) * pc.kh * pc.kw * ho * wo * exe
        prof.bytes += 4.0 * (n * c * h * w + pc.cout * (pc.cin // pc.groups) * pc.kh * pc.kw
                             + n * pc.cout * ho * wo * (1 + (r4 is not None) + bool(accumulate)))
        prof.launches += 1
        prof.shapes.append(("N%d C%d>%d %dx%d k%dx%d s%d,%d d%d,%d g%d%s%s%s" % (
            n, c, pc.cout, h, w, pc.kh, pc.kw, pc.stride[0], pc.stride[1], pc.dilation[0], pc.dilation[1], pc.groups,
            " res" if r4 is not None else "", " acc" if accumulate else "", (" shuf" if shuffle else "") + (" wino2" if wino2 else " wino" if wino else " wino1d" if wino1 else "")),
            2.0 * n * pc.cout * (pc.cin // pc.groups) * pc.kh * pc.kw * ho * wo))
    return out


# ---------------------------------------------------------------------------------------------------
# Transposed convolution = 1x1 GEMM (aicg_conv_forward) + gather (aicg_col2im)
# ---------------------------------------------------------------------------------------------------
class PackedConvTranspose:
    """ConvTranspose1d/2d weights (Cin, Cout, K) / (Cin, Cout, KH, KW) re-laid-out as the (Cout*KH*KW, Cin)
    matrix of the 1x1 GEMM; `bias` is applied by col2im."""

    def __init__(self, weight, bias=None, stride=1, padding=0, output_padding=0, device=None):
        if weight.dim() == 3:
            weight = weight.unsqueeze(2)
            stride, padding, output_padding = (1, _one(stride)), (0, _one(padding)), (0, _one(output_padding))
        else:
            stride, padding, output_padding = _pair(stride), _pair(padding), _pair(output_padding)
        self.cin, self.cout, self.kh, self.kw = weight.shape
        self.stride, self.padding, self.output_padding = stride, padding, output_padding
        device = weight.device if device is None else device
        gemm_w = weight.detach().permute(1, 2, 3, 0).reshape(self.cout * self.kh * self.kw, self.cin)
        self.gemm = PackedConv(gemm_w.contiguous(), None, device=device)
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        # per-GEMM-row bias for the fused kernel = stride = 2 form (row = co * 4 + tap)
        self.bias4 = None if self.bias is None else self.bias.repeat_interleave(self.kh * self.kw).contiguous()

    def out_hw(self, h, w):
        ho = (h - 1) * self.stride[0] - 2 * self.padding[0] + self.kh + self.output_padding[0]
        wo = (w - 1) * self.stride[1] - 2 * self.padding[1] + self.kw + self.output_padding[1]
        return ho, wo


def conv_transpose(x, pt, out=None, add=None, pre_act=ACT_NONE, pre_slope=0.0, act=ACT_NONE, act_slope=0.0, mul=None):
    """y = act(conv_transpose(pre_act(x)) + bias) + add   (or ... * mul: the MDX U-Net's multiplicative skip)."""
    is1d = x.dim() == 3
    x4 = _as4d(x)
    n, c, h, w = x4.shape
    assert add is None or mul is None
    if (not is1d and (pt.kh, pt.kw) == (2, 2) and pt.stride == (2, 2) and pt.padding == (0, 0)
            and pt.output_padding == (0, 0)):
        # non-overlapping taps: every output element is one GEMM element -> scatter + bias + act + skip in the GEMM epilogue
        return conv(x4, pt.gemm, res=add if mul is None else mul, out=out, pre_act=pre_act, pre_slope=pre_slope, act=act,
                    act_slope=act_slope, bias=pt.bias4, shuffle=2, res_mul=mul is not None)
    cols = conv(x4, pt.gemm, pre_act=pre_act, pre_slope=pre_slope)  # (N, Cout*KH*KW, H, W)
    ho, wo = pt.out_hw(h, w)
    if out is None:
        out = torch.empty((n, pt.cout, wo) if is1d else (n, pt.cout, ho, wo), dtype=torch.float32, device=x.device)
    o4 = _as4d(out)
    assert o4.shape == (n, pt.cout, ho, wo) and (o4.stride(3) == 1 or wo == 1)
    a4 = None if add is None else _as4d(add)
    if a4 is not None:
        assert a4.shape == o4.shape and (a4.stride(3) == 1 or wo == 1)
    _check(cols, out, add, pt.bias)
    asn, asc, ash = (a4.stride(0), a4.stride(1), a4.stride(2)) if a4 is not None else (0, 0, 0)
    _call("aicg_col2im", _ptr(cols), _ptr(pt.bias), _ptr(a4), _ptr(o4), n, pt.cout, h, w, ho, wo, pt.kh, pt.kw,
              pt.stride[0], pt.stride[1], pt.padding[0], pt.padding[1], act, act_slope,
              o4.stride(0), o4.stride(1), o4.stride(2), asn, asc, ash, _stream(x))
    if mul is not None:
        globals()["mul"](out, mul, out=out)
    return out


# ---------------------------------------------------------------------------------------------------
# Synthesizer helpers
# ---------------------------------------------------------------------------------------------------
def sine_source(f0, noise, upp, sr, lin_w, lin_b, sine_amp=0.1, noise_std=0.003):
    """f0: (T,) Hz; noise: (T*upp,) N(0,1) draws -> harmonic source (T*upp,) after l_linear + tanh."""
    f0 = f0.contiguous().float()
    noise = noise.contiguous().float()
    t = f0.numel()
    assert noise.numel() == t * upp
    _check(f0, noise)
    prefix = torch.empty(t, dtype=torch.float64, device=f0.device)
    out = torch.empty(t * upp, dtype=torch.float32, device=f0.device)
    _call("aicg_sine_source", _ptr(f0), _ptr(noise), _ptr(prefix), _ptr(out), t, int(upp), float(sr), sine_amp,
              noise_std, float(lin_w), float(lin_b), _stream(f0))
    return out


def gate_tanh_sigmoid(a, out=None):
    """a: (N, 2C, T) contiguous -> tanh(a[:, :C]) * sigmoid(a[:, C:])."""
    assert a.is_contiguous() and a.dim() == 3 and a.shape[1] % 2 == 0
    n, c2, t = a.shape
    if out is None:
        out = torch.empty((n, c2 // 2, t), dtype=torch.float32, device=a.device)
    assert out.is_contiguous()
    _check(a, out)
    _call("aicg_gate_tanh_sigmoid", _ptr(a), _ptr(out), n, c2 // 2, t, _stream(a))
    return out


def prior_sample(stats, noise, scale):
    """stats: (1, 2C, T) = [m; logs]; noise (1, C, T) -> m + exp(logs) * noise * scale."""
    assert stats.is_contiguous() and noise.is_contiguous() and stats.shape[0] == 1
    c, t = noise.shape[1], noise.shape[2]
    assert stats.shape[1] == 2 * c and stats.shape[2] == t
    out = torch.empty_like(noise)
    _check(stats, noise)
    _call("aicg_prior_sample", _ptr(stats), _ptr(noise), _ptr(out), c, t, float(scale), _stream(stats))
    return out


def feats_prepare(feats, t_out, feats0=None, pitchf=None, protect=0.5):
    """(Th, C) token-major features -> (1, C, t_out) channel-major, nearest x2 upsampled, protect-blended."""
    feats = feats.contiguous()
    th, c = feats.shape
    out = torch.empty((1, c, t_out), dtype=torch.float32, device=feats.device)
    if feats0 is not None:
        feats0 = feats0.contiguous()
        pitchf = pitchf.contiguous().float()
        assert pitchf.numel() >= t_out
    _check(feats, feats0, pitchf)
    _call("aicg_feats_prepare", _ptr(feats), _ptr(feats0), _ptr(pitchf), _ptr(out), th, c, int(t_out), float(protect),
              _stream(feats))
    return out


def layernorm_ct(x, gamma, beta, res=None, out=None, eps=1e-5):
    """LayerNorm over channels of contiguous (N, C, T); out = LN(x + res)."""
    assert x.is_contiguous() and x.dim() == 3
    n, c, t = x.shape
    if out is None:
        out = torch.empty_like(x)
    if res is not None:
        assert res.is_contiguous() and res.shape == x.shape
    _check(x, res, gamma, beta, out)
    _call("aicg_layernorm_ct", _ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(out), n, c, t, float(eps), c * t, c * t,
              c * t, _stream(x))
    return out


def rownorm_act(x, gamma, beta, act=ACT_NONE, eps=1e-5, out=None):
    """x: (rows, T), unit stride along T: per-row (x - mean)/sqrt(var + eps) * gamma + beta, then act.  x (and out) may be views of rows
    padded to a common stride (x.stride(0) >= T); `out` defaults to a buffer laid out like x."""
    assert x.dim() == 2 and (x.stride(1) == 1 or x.shape[1] == 1)
    rows, t = x.shape
    ld = x.stride(0) if rows > 1 else max(x.stride(0), t)
    if out is None:
        out = torch.empty((rows, ld), dtype=torch.float32, device=x.device)[:, :t] if ld != t else torch.empty_like(x)
    assert out.shape == x.shape and (out.stride(1) == 1 or t == 1) and (rows <= 1 or out.stride(0) == ld)
    _check(x, gamma, beta, out)
    need = ctypes.c_int64(0)
    _call("aicg_rownorm_act_workspace_floats", rows, t, ctypes.addressof(need))
    ws = torch.empty(need.value, dtype=torch.float32, device=x.device) if need.value else None
    _call("aicg_rownorm_act_ld", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), rows, t, ld, float(eps), act, _ptr(ws), _stream(x))
    return out


def attention(q, k, v, n_heads, relk=None, relv_emb=None, window=0, scale=1.0, n_splits=None):
    """q, k, v: (C, T) channel-major (rows may be slices of a fused QKV buffer; stride(1) == 1).
    relk: (H, 2w+1, T) precomputed q.E^k (or None); relv_emb: (2w+1, D) for the relative value term."""
    c, t = q.shape
    d = c // n_heads
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
    o = torch.empty((c, t), dtype=torch.float32, device=q.device)
    lse = torch.empty((n_heads, t), dtype=torch.float32, device=q.device) if relv_emb is not None else None
    if relk is not None:
        assert relk.is_contiguous() and relk.shape == (n_heads, 2 * window + 1, t)
    _check(q, k, v, relk, relv_emb)
    st = _stream(q)
    # (query block, head) pairs alone do not fill 256 CUs for a few heads: split the keys until ~1024 workgroups exist
    blocks = -(-t // 128) * n_heads
    # (> 4: the merge pass costs more than it fills; HuBERT's 312 blocks, round-5 kernel: 4 splits 0.353 ms, 3 0.361, 2 0.398, 8 0.377)
    splits = max(1, min(4, -(-t // 32), -(-1024 // blocks))) if n_splits is None else n_splits
    if splits > 1:
        scratch = torch.empty(splits * n_heads * (d + 2) * t, dtype=torch.float32, device=q.device)
        _call("aicg_attention_split", _ptr(q), _ptr(k), _ptr(v), _ptr(relk), _ptr(o), _ptr(lse), t, n_heads, d, window,
                  q.stride(0), k.stride(0), v.stride(0), o.stride(0), float(scale), splits, _ptr(scratch), st)
    else:
        _call("aicg_attention", _ptr(q), _ptr(k), _ptr(v), _ptr(relk), _ptr(o), _ptr(lse), t, n_heads, d, window,
                  q.stride(0), k.stride(0), v.stride(0), o.stride(0), float(scale), st)
    if relv_emb is not None:
        relv_emb = relv_emb.contiguous()
        _call("aicg_attention_relv", _ptr(q), _ptr(k), _ptr(relk), _ptr(relv_emb), _ptr(lse), _ptr(o), t, n_heads, d,
                  window, q.stride(0), k.stride(0), o.stride(0), float(scale), st)
    return o


# ---------------------------------------------------------------------------------------------------
# RMVPE helpers
# ---------------------------------------------------------------------------------------------------
ACT_LOGCLAMP = 6


def complex_abs(re, im):
    assert re.is_contiguous() and im.is_contiguous() and re.shape == im.shape
    out = torch.empty_like(re)
    _check(re, im)
    _call("aicg_complex_abs", _ptr(re), _ptr(im), _ptr(out), re.numel(), _stream(re))
    return out


def channel_affine(x, scale, shift, act=ACT_NONE):
    assert x.is_contiguous() and x.dim() >= 2
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    out = torch.empty_like(x)
    _check(x, scale, shift)
    _call("aicg_channel_affine", _ptr(x), _ptr(scale), _ptr(shift), _ptr(out), n, c, hw, act, _stream(x))
    return out


def avgpool2x2(x):
    n, c, h, w = x.shape
    assert x.stride(3) == 1
    out = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    _check(x)
    _call("aicg_avgpool2x2", _ptr(x), _ptr(out), n, c, h, w, x.stride(0), x.stride(1), x.stride(2), _stream(x))
    return out


GRU_TWO_WORKGROUPS = _env.dev("AICG_GRU_2WG", "1") != "0"     # "0": the single-workgroup kernel (no co-residency needed)
GRU_WORKGROUPS = int(_env.dev("AICG_GRU_WG", "4"))             # workgroups per direction of the multi-workgroup form: 4 or 2


_gru_pending = []


def gru_timed_out():
    """True if any multi-workgroup GRU launch since the last call reported a partner-exchange timeout (the partner workgroup
    was not co-resident in time: a busy or shared GPU).  Call after the stream that ran them has been drained.  The result
    of such a launch is invalid; callers recompute with the single-workgroup kernel (`gru_bidir(two_workgroups=False)`)."""
    pend, _gru_pending[:] = list(_gru_pending), []     # (the tensor references of the launches go with the list)
    bad = False
    for f, _ in pend:
        bad = bad or int(f.item()) != 0
    return bad


def gru_check_pending():
    """Raise if a multi-workgroup GRU launch timed out (for callers that cannot recompute)."""
    if gru_timed_out():
        raise RuntimeError("aicg_gru_bidir_2wg: partner workgroup exchange timed out")


def _gru_repair_backlog():
    """A caller that never polls gru_timed_out(): bound the backlog (this synchronises) and REPAIR instead of failing -- every pending
    launch whose flag is set is recomputed by the single-workgroup kernel into the same output tensor, on the stream that is current
    NOW.  The repair fixes `out` for LATER readers only: anything already derived from a flagged output (classifier, decode) stays
    wrong and is not re-run -- which is why a flagged launch is also reported (warning), and why callers that can recompute their
    dependants should poll gru_timed_out() themselves, as pipeline() and RMVPE.infer_from_audio do."""
    import warnings
    pend, _gru_pending[:] = list(_gru_pending), []
    for f, (gi, whh_t, bhh, out, hidden) in pend:
        if int(f.item()) != 0:
            warnings.warn("aicg_gru_bidir: a multi-workgroup launch timed out %d launches ago and nobody polled gru_timed_out(); its "
                          "output is recomputed in place now -- results already derived from it are stale" % len(pend))
            _call("aicg_gru_bidir", _ptr(gi), _ptr(whh_t), _ptr(bhh), _ptr(out), hidden, gi.shape[1], _stream(gi))


def gru_bidir(gi, whh_t, bhh, hidden, two_workgroups=None):
    """gi: (6*hidden, T) channel-major input projections -> (2*hidden, T).  Default: the multi-workgroup-per-direction
    kernel (all of W_hh on chip); `two_workgroups=False` (or AICG_GRU_2WG=0) selects the single-workgroup kernel.
    The multi-workgroup kernels need their workgroups co-resident; a launch that timed out waiting for a partner sets a flag that
    pipeline() polls with gru_timed_out() (and recomputes).  Callers that never poll are covered too: once 16 launches are
    pending, the flagged ones are recomputed in place by the single-workgroup kernel (_gru_repair_backlog)."""
    assert gi.is_contiguous() and gi.shape[0] == 6 * hidden
    t = gi.shape[1]
    out = torch.empty((2 * hidden, t), dtype=torch.float32, device=gi.device)
    _check(gi, whh_t, bhh)
    if GRU_TWO_WORKGROUPS if two_workgroups is None else two_workgroups:
        scratch = torch.empty(32 * hidden + 64, dtype=torch.uint8, device=gi.device)
        # hidden 256 (every RMVPE): four workgroups per direction, all of W_hh in registers; other sizes: two (part of it in LDS)
        _call("aicg_gru_bidir_4wg" if hidden == 256 and GRU_WORKGROUPS == 4 else "aicg_gru_bidir_2wg", _ptr(gi), _ptr(whh_t), _ptr(bhh),
              _ptr(out), hidden, t, _ptr(scratch), _stream(gi))
        # the kernel's exchange-timeout flag is read back lazily (gru_timed_out): an .item() here would park the host
        # until the recurrence ends, which is exactly the time pipeline() wants to spend queueing HuBERT work
        _gru_pending.append((scratch[32 * hidden: 32 * hidden + 4].view(torch.int32), (gi, whh_t, bhh, out, hidden)))
        if len(_gru_pending) > 16:
            _gru_repair_backlog()
        return out
    _call("aicg_gru_bidir", _ptr(gi), _ptr(whh_t), _ptr(bhh), _ptr(out), hidden, t, _stream(gi))
    return out


class GruSegments:
    """The BiGRU recurrence of gru_bidir cut into segments of steps (aicg_gru_bidir*_seg): `run(s0, s1)` queues steps s0 .. s1 - 1 of
    both directions -- forward frames s0 .. s1 - 1, backward frames T - s1 .. T - s0 - 1 -- on the current stream.  Bit-identical to the
    one-launch form (the hidden state is carried in fp32); what it buys is that the frames BOTH directions have passed,
    `ready(s1)` = [T - s1, s1), grow from the middle of the track while the recurrence is still on its way to the ends."""

    def __init__(self, gi, whh_t, bhh, hidden, two_workgroups=None):
        assert gi.is_contiguous() and gi.shape[0] == 6 * hidden
        _check(gi, whh_t, bhh)
        self.gi, self.whh_t, self.bhh, self.hidden, self.T = gi, whh_t, bhh, hidden, gi.shape[1]
        self.out = torch.empty((2 * hidden, self.T), dtype=torch.float32, device=gi.device)
        self.state = torch.zeros((2, hidden), dtype=torch.float32, device=gi.device)
        multi = GRU_TWO_WORKGROUPS if two_workgroups is None else two_workgroups
        self.multi = bool(multi) and hidden == 256 and GRU_WORKGROUPS == 4     # the four-workgroup kernel; otherwise the single-workgroup one
        self.scratch = torch.empty(32 * hidden + 64, dtype=torch.uint8, device=gi.device) if self.multi else None
        self.done = 0
        # the error word inside `scratch` is zeroed by the FIRST segment's launch (on the stream that launch runs on): until that launch
        # has completed the word is uninitialised allocator memory, and timed_out() -- polled from another stream -- must not read it
        self._first_done = None

    def run(self, s1):
        """Queue steps self.done .. s1 - 1."""
        s0, s1 = self.done, min(int(s1), self.T)
        if s1 <= s0:
            return
        if self.multi:
            _call("aicg_gru_bidir_4wg_seg", _ptr(self.gi), _ptr(self.whh_t), _ptr(self.bhh), _ptr(self.out), self.hidden, self.T, s0, s1,
                  _ptr(self.state), _ptr(self.scratch), _stream(self.gi))
            if s0 == 0 and self.gi.is_cuda:
                self._first_done = torch.cuda.Event()
                self._first_done.record(torch.cuda.current_stream(self.gi.device))
            if s1 == self.T:   # one flag for the whole recurrence (the error word accumulates over the segments)
                _gru_pending.append((self.scratch[32 * self.hidden: 32 * self.hidden + 4].view(torch.int32),
                                     (self.gi, self.whh_t, self.bhh, self.out, self.hidden)))
        else:
            _call("aicg_gru_bidir_seg", _ptr(self.gi), _ptr(self.whh_t), _ptr(self.bhh), _ptr(self.out), self.hidden, self.T, s0, s1,
                  _ptr(self.state), _stream(self.gi))
        self.done = s1

    def timed_out(self):
        """The recurrence's error word as of NOW (a 4-byte read on the current stream; the segments may still be running on another):
        nonzero = a partner exchange of the multi-workgroup kernel timed out, the output of that and every later segment is invalid
        (later segments bail out early).  pipeline() polls it between chunks instead of synthesising a whole track from bad pitch.
        False while the first segment -- whose launch zeroes the word -- has not run to completion (nothing can have timed out yet, and
        the word is not initialised before it: ADVICE r5)."""
        if not self.multi or self.done == 0:
            return False
        if self._first_done is not None and not self._first_done.query():
            return False
        return int(self.scratch[32 * self.hidden: 32 * self.hidden + 4].view(torch.int32).item()) != 0

    def ready(self):
        """[lo, hi): frames whose forward AND backward state exist after the steps queued so far (empty: lo >= hi)."""
        return max(0, self.T - self.done), min(self.T, self.done)


def salience_decode(salience, thred=0.03, want_center=False):
    """salience (T, 360) fp32 row-major -> (cents float64 (T,), f0 float64 (T,)[, argmax int32 (T,)])."""
    salience = salience.contiguous().float()
    t, nb = salience.shape
    cents = torch.empty(t, dtype=torch.float64, device=salience.device)
    f0 = torch.empty(t, dtype=torch.float64, device=salience.device)
    center = torch.empty(t, dtype=torch.int32, device=salience.device) if want_center else None
    _check(salience)
    _call("aicg_salience_decode", _ptr(salience), _ptr(cents), _ptr(f0), _ptr(center), t, nb, float(thred), _stream(salience))
    return (cents, f0, center) if want_center else (cents, f0)


def f0_coarse(f0, factor, mel_min, mel_max):
    """f0 float64 (n,) -> (f0 * factor float64, coarse int64 in [1, 255])."""
    f0 = f0.contiguous().double()
    out = torch.empty_like(f0)
    coarse = torch.empty(f0.shape, dtype=torch.int64, device=f0.device)
    _check(f0)
    _call("aicg_f0_coarse", _ptr(f0), float(factor), _ptr(out), _ptr(coarse), f0.numel(), float(mel_min), float(mel_max),
              _stream(f0))
    return out, coarse


# ---------------------------------------------------------------------------------------------------
# MDX-Net helpers
# ---------------------------------------------------------------------------------------------------
def linear_last(x, weight, bias=None, ch_scale=None, ch_shift=None, act=ACT_NONE, res=None, out=None, fp32=False):
    """nn.Linear over the last axis of a contiguous (B, C, T, F) map (+ per-channel affine, act, residual).  `fp32`: never the
    split-precision kernel (layers that select indices: CREPE)."""
    assert x.is_contiguous() and x.dim() == 4 and weight.is_contiguous()
    b, c, t, f = x.shape
    o = weight.shape[0]
    assert weight.shape[1] == f
    if out is None:
        out = torch.empty((b, c, t, o), dtype=torch.float32, device=x.device)
    if res is not None:
        assert res.is_contiguous() and res.shape == out.shape
    _check(x, weight, bias, ch_scale, ch_shift, res, out)
    _call("aicg_gemm_nt_split" if split_precision and not fp32 else "aicg_gemm_nt", _ptr(x), _ptr(weight), _ptr(bias), _ptr(ch_scale),
          _ptr(ch_shift), _ptr(res), _ptr(out), b * c * t, f, o, f, f, o, o, t, c, act, _stream(x))
    return out


def pack_tdf_w1(w1):
    """(H, F) -> [F / 8][2][H][4]: element e of quad (g, par, h) = W1[h][8 g + 2 e + par] (aicg_tdf_pair's w1_packed)."""
    h, f = w1.shape
    assert f % 8 == 0
    return w1.detach().float().t().reshape(f // 8, 4, 2, h).permute(0, 2, 3, 1).contiguous()


def pack_tdf_w2(w2):
    """(F, H) -> [F / 32] slabs of 32 rows x (H + 4) floats, each slab zero-padded to a multiple of 256 floats (aicg_tdf_pair's
    w2_packed: the LDS image of a stage, row padding included, so that a stage is one contiguous HBM -> LDS DMA)."""
    f, h = w2.shape
    assert f % 32 == 0
    slab = (32 * (h + 4) + 255) // 256 * 256
    out = torch.zeros((f // 32, slab), dtype=torch.float32, device=w2.device)
    rows = torch.zeros((f // 32, 32, h + 4), dtype=torch.float32, device=w2.device)
    rows[:, :, :h] = w2.detach().float().reshape(f // 32, 32, h)
    out[:, : 32 * (h + 4)] = rows.reshape(f // 32, -1)
    return out.contiguous()


def tdf_pair_supported(f, h, rows_per_ch):
    return bool(_lib.get().aicg_tdf_pair_supported(int(f), int(h), int(rows_per_ch)))


def tdf_pair(x, w1p, b1, s1, t1, w2p, b2, s2, t2, out=None):
    """x + relu(bn2(relu(bn1(x W1^T + b1)) W2^T + b2)) over the last axis of a contiguous (B, C, T, F) map, one launch.
    w1p = pack_tdf_w1(W1), w2p = pack_tdf_w2(W2)."""
    assert x.is_contiguous() and x.dim() == 4 and w2p.is_contiguous() and w1p.is_contiguous()
    b, c, t, f = x.shape
    h = w1p.numel() // f
    assert w1p.numel() == f * h and w2p.shape[0] == f // 32
    if out is None:
        out = torch.empty_like(x)
    _check(x, w1p, b1, s1, t1, w2p, b2, s2, t2, out)
    _call("aicg_tdf_pair", _ptr(x), _ptr(w1p), _ptr(b1), _ptr(s1), _ptr(t1), _ptr(w2p), _ptr(b2), _ptr(s2), _ptr(t2), _ptr(out),
          b * c * t, f, h, t, c, _stream(x))
    return out


def mul(a, b, out=None):
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    if out is None:
        out = torch.empty_like(a)
    _check(a, b, out)
    _call("aicg_mul", _ptr(a), _ptr(b), _ptr(out), a.numel(), _stream(a))
    return out


def axpbypcz(a, alpha, b=None, beta=0.0, c=None, gamma=0.0, out=None):
    """out = alpha*a + beta*b + gamma*c over contiguous fp32 tensors of one shape."""
    assert a.is_contiguous() and (b is None or b.is_contiguous()) and (c is None or c.is_contiguous())
    if out is None:
        out = torch.empty_like(a)
    _check(a, b, c, out)
    _call("aicg_axpbypcz", _ptr(a), float(alpha), _ptr(b), float(beta), _ptr(c), float(gamma), _ptr(out), a.numel(),
              _stream(a))
    return out


# ---------------------------------------------------------------------------------------------------
# VC.pipeline pre/post on the device
# ---------------------------------------------------------------------------------------------------
def box_sum_f64(x, n, window):
    """x: float64 (n + window - 1,) -> out[j] = sum_{i<window} x[j+i] (reference summation order)."""
    x = x.contiguous()
    assert x.dtype == torch.float64 and x.numel() >= n + window - 1
    out = torch.empty(n, dtype=torch.float64, device=x.device)
    _check(x)
    _call("aicg_box_sum_f64", _ptr(x), _ptr(out), n, window, _stream(x))
    return out


def argmin_abs_f64(x, starts, lens):
    """first index of min |x[s : s+l]| for each (s, l)."""
    assert x.dtype == torch.float64
    st = torch.tensor(list(starts), dtype=torch.int64, device=x.device)
    ln = torch.tensor(list(lens), dtype=torch.int64, device=x.device)
    out = torch.empty(len(st), dtype=torch.int64, device=x.device)
    _check(x)
    _call("aicg_argmin_abs_f64", _ptr(x), _ptr(st), _ptr(ln), _ptr(out), len(st), _stream(x))
    return out


def frame_rms(x, frame_length, hop_length):
    """librosa.feature.rms (center, reflect) of a 1-D float32 / float64 device tensor -> float64 (n_frames,)."""
    x = x.contiguous()
    assert x.dim() == 1 and x.dtype in (torch.float32, torch.float64)
    n = x.numel()
    out = torch.empty(1 + n // hop_length, dtype=torch.float64, device=x.device)
    _check(x)
    _call("aicg_frame_rms", _ptr(x), 1 if x.dtype == torch.float64 else 0, _ptr(out), n, frame_length, hop_length, _stream(x))
    return out


def rms_mix_(data, rms1, rms2, rate):
    assert data.is_contiguous() and data.dtype == torch.float32
    _check(data, rms1, rms2)
    _call("aicg_rms_mix", _ptr(data), data.numel(), _ptr(rms1), rms1.numel(), _ptr(rms2), rms2.numel(), float(rate),
              _stream(data))
    return data


def absmax(x):
    assert x.is_contiguous() and x.dtype == torch.float32
    out = torch.zeros(1, dtype=torch.float32, device=x.device)
    _check(x)
    _call("aicg_absmax", _ptr(x), x.numel(), _ptr(out), _stream(x))
    return out


def to_int16(x, scale):
    assert x.is_contiguous() and x.dtype == torch.float32
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    _check(x)
    _call("aicg_to_int16", _ptr(x), _ptr(out), x.numel(), float(scale), _stream(x))
    return out


# ---------------------------------------------------------------------------------------------------
# CREPE helpers
# ---------------------------------------------------------------------------------------------------
def frame_normalize(frames):
    assert frames.is_contiguous() and frames.dim() == 2
    out = torch.empty_like(frames)
    _check(frames)
    _call("aicg_frame_normalize", _ptr(frames), _ptr(out), frames.shape[0], frames.shape[1], _stream(frames))
    return out


def affine_maxpool2(x, scale, shift):
    """x (N, C, W) contiguous -> max over pairs of (x * scale[c] + shift[c]) -> (N, C, W/2)."""
    assert x.is_contiguous() and x.dim() == 3
    n, c, w = x.shape
    out = torch.empty((n, c, w // 2), dtype=torch.float32, device=x.device)
    _check(x, scale, shift)
    _call("aicg_affine_maxpool2", _ptr(x), _ptr(scale), _ptr(shift), _ptr(out), n, c, w, _stream(x))
    return out


def crepe_viterbi(probs, seq_len, bin_lo, bin_hi):
    """probs (n_seq, 360, max_steps) fp32, seq_len list -> bins int64 (n_seq, max_steps)."""
    probs = probs.contiguous()
    n_seq, nb, ms = probs.shape
    sl = torch.tensor(list(seq_len), dtype=torch.int32, device=probs.device)
    logp = torch.empty((n_seq, ms, nb), dtype=torch.float32, device=probs.device)
    ptr = torch.empty((n_seq, ms, nb), dtype=torch.int16, device=probs.device)
    bins = torch.zeros((n_seq, ms), dtype=torch.int64, device=probs.device)
    _check(probs)
    _call("aicg_crepe_viterbi", _ptr(probs), _ptr(sl), _ptr(logp), _ptr(ptr), _ptr(bins), n_seq, nb, ms, int(bin_lo),
              int(bin_hi), _stream(probs))
    return bins


# ---------------------------------------------------------------------------------------------------
# Signal processing either side of the networks (csrc/dsp.hip)
# ---------------------------------------------------------------------------------------------------
def filter3(x, mode):
    """3-tap NaN-aware filter of a float32 frame sequence: mode 'median' (lower median) or 'mean'
    (torchcrepe.filter.median / .mean with win_length 3)."""
    x = x.contiguous().float()
    out = torch.empty_like(x)
    _check(x)
    _call("aicg_filter3", _ptr(x), _ptr(out), x.numel(), {"median": 0, "mean": 1}[mode], _stream(x))
    return out


_filtfilt_plans = {}


def _filtfilt_plan(b, a):
    """Host-side constants of scipy.signal.filtfilt(b, a, .) with default arguments: normalised coefficients, lfilter_zi initial
    conditions, padlen, and the warm-up length after which a zero-state start is indistinguishable in float64."""
    key = (tuple(np.asarray(b, np.float64)), tuple(np.asarray(a, np.float64)))
    if key not in _filtfilt_plans:
        from scipy.signal import lfilter_zi
        bb, aa = np.asarray(b, np.float64), np.asarray(a, np.float64)
        n = max(len(aa), len(bb))
        bb = np.concatenate([bb, np.zeros(n - len(bb))])
        aa = np.concatenate([aa, np.zeros(n - len(aa))])
        zi = np.ascontiguousarray(lfilter_zi(bb, aa), dtype=np.float64)
        rmax = float(np.max(np.abs(np.roots(aa)))) if n > 1 else 0.0
        if not rmax < 1.0:
            raise ValueError("filtfilt: unstable filter (max |pole| = %g)" % rmax)
        # warm-up: where the envelope of the impulse response has fallen 18 decades (clustered poles -- a Butterworth's -- decay
        # like n^k r^n, so the pole radius alone under-estimates it)
        from scipy.signal import lfilter
        imp = np.zeros(1 << 17)
        imp[0] = 1.0
        env = np.maximum.accumulate(np.abs(lfilter(bb, aa, imp))[::-1])[::-1]
        below = np.nonzero(env < 1e-18 * env[0])[0]
        if len(below) == 0:
            raise ValueError("filtfilt: impulse response too long for the block-parallel recurrence")
        warm = (int(below[0]) + 255) // 256 * 256
        _filtfilt_plans[key] = (np.ascontiguousarray(bb), np.ascontiguousarray(aa), zi, n - 1, 3 * n, warm)
    return _filtfilt_plans[key]


def filtfilt_f64(x, b, a, block=1024):
    """scipy.signal.filtfilt(b, a, x) (default odd padding / lfilter_zi) of a float64 device signal, block-parallel."""
    x = x.contiguous().double()
    bb, aa, zi, order, padlen, warm = _filtfilt_plan(b, a)
    n = x.numel()
    y = torch.empty_like(x)
    ext = torch.empty(n + 2 * padlen, dtype=torch.float64, device=x.device)
    mid = torch.empty_like(ext)
    _check(x)
    _call("aicg_filtfilt_f64", _ptr(x), _ptr(y), n, bb.ctypes.data, aa.ctypes.data, zi.ctypes.data, order, padlen, int(block),
          int(warm), _ptr(ext), _ptr(mid), _stream(x))
    return y


_resample_plans = {}


def _resample_plan(up, down, n_in, device):
    """Filter and trimming of scipy.signal.resample_poly(x float32, up, down) (Kaiser-5 windowed sinc, 10 * max(up, down) zero
    crossings each side), as the polyphase table the kernel reads."""
    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    key = (up, down, str(device))
    if key not in _resample_plans:
        from scipy.signal import firwin
        max_rate = max(up, down)
        half_len = 10 * max_rate
        h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)).astype(np.float32)
        h *= up
        n_pre_pad = down - half_len % down
        hpad = np.concatenate([np.zeros(n_pre_pad, np.float32), h])
        taps = (len(hpad) + up - 1) // up
        table = np.zeros((up, taps), np.float32)
        for ph in range(up):
            col = hpad[ph::up]
            table[ph, :len(col)] = col
        _resample_plans[key] = (torch.from_numpy(table).to(device), taps, (half_len + n_pre_pad) // down)
    table, taps, pre = _resample_plans[key]
    n_out = n_in * up
    n_out = n_out // down + bool(n_out % down)
    return up, down, table, taps, pre, n_out


def resample_poly_mono(x, sr_in, sr_out):
    """(C, N) or (N,) float32 device signal at sr_in -> (n_out,) float32 mono at sr_out: channel mean + polyphase FIR with
    scipy.signal.resample_poly's filter and trimming."""
    x = x.float()
    if x.dim() == 1:
        x = x.unsqueeze(0)
    assert x.stride(1) == 1
    c, n = x.shape
    up, down, table, taps, pre, n_out = _resample_plan(sr_out, sr_in, n, x.device)
    y = torch.empty(n_out, dtype=torch.float32, device=x.device)
    _check(x, table)
    _call("aicg_resample_poly", _ptr(x), _ptr(y), n, n_out, c, x.stride(0), up, down, _ptr(table), taps, pre, _stream(x))
    return y


def row_sqnorm(v):
    v = v.contiguous().float()
    out = torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
    _check(v)
    _call("aicg_row_sqnorm", _ptr(v), _ptr(out), v.shape[0], v.shape[1], _stream(v))
    return out


def knn8_update(dots, xnorm, qnorm, col_off, best_d, best_i, merge):
    """dots (rows, cols) inner products of one column chunk of the index -> running 8 nearest (squared L2, ascending)."""
    assert dots.dim() == 2 and dots.stride(1) == 1
    _check(dots, xnorm, qnorm, best_d, best_i)
    _call("aicg_knn8", _ptr(dots), dots.stride(0), _ptr(xnorm), _ptr(qnorm), dots.shape[0], dots.shape[1], int(col_off),
          _ptr(best_d), _ptr(best_i), 1 if merge else 0, _stream(dots))


def ivf_scan8(feats, vecs, list_off, probe, nprobe):
    """IndexIVFFlat scan: feats (rows, dim), vecs (N, dim) stored list by list, list_off (nlist + 1,) int64, probe (rows, >= nprobe)
    int64 nearest-first list ids -> (squared distances (rows, 8) ascending, storage positions (rows, 8), -1 where none)."""
    assert feats.is_contiguous() and vecs.is_contiguous() and probe.is_contiguous() and list_off.dtype == torch.int64
    rows, dim = feats.shape
    best_d = torch.empty((rows, 8), dtype=torch.float32, device=feats.device)
    best_p = torch.empty((rows, 8), dtype=torch.int64, device=feats.device)
    _check(feats, vecs, list_off, probe)
    _call("aicg_ivf_scan8", _ptr(feats), _ptr(vecs), _ptr(list_off), _ptr(probe), probe.shape[1], int(nprobe), rows, dim,
          _ptr(best_d), _ptr(best_p), _stream(feats))
    return best_d, best_p


def index_mix_(feats, big, best_d, best_i, rate, recompute=False):
    """feats (rows, dim) <- rate * sum_k w_k big[best_i[k]] + (1 - rate) * feats, w = inverse-square-distance weights (best_i < 0:
    weight 0).  `recompute`: distances re-evaluated directly from the gathered rows first (and stored in best_d)."""
    assert feats.is_contiguous() and big.is_contiguous() and best_d.is_contiguous() and best_i.is_contiguous()
    _check(feats, big, best_d, best_i)
    _call("aicg_index_mix", _ptr(feats), _ptr(big), _ptr(best_d), _ptr(best_i), feats.shape[0], feats.shape[1], float(rate),
          1 if recompute else 0, _stream(feats))
    return feats


# ---------------------------------------------------------------------------------------------------
# Optional per-stage accounting for bench.py's `stages` list (HIP events on the launch stream around the wrapped ops)
# ---------------------------------------------------------------------------------------------------
class StageProfile:
    """name -> launches, algorithmic flops / bytes, HIP-event time.  Enabled with `ops.stage_profile = StageProfile()`."""

    def __init__(self):
        self.rec = {}

    def add(self, name, e0, e1, flops, nbytes):
        r = self.rec.setdefault(name, {"events": [], "flops": 0.0, "bytes": 0.0, "launches": 0})
        r["events"].append((e0, e1))
        r["flops"] += flops
        r["bytes"] += nbytes
        r["launches"] += 1

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, r in self.rec.items():
            ms = sum(a.elapsed_time(b) for a, b in r["events"])
            out[name] = {"launches": r["launches"], "ms": ms, "flops": r["flops"], "bytes": r["bytes"]}
        return out


stage_profile = None


def _staged(name, fn, work):
    """work(args, kwargs, result) -> (algorithmic flops, algorithmic HBM bytes) of one call."""
    def wrapped(*a, **k):
        prof = stage_profile
        t = a[0] if a and torch.is_tensor(a[0]) else None
        if prof is None or t is None or not t.is_cuda:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        fl, by = work(a, k, r)
        prof.add(name, e0, e1, fl, by)
        return r
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


def _numel(*ts):
    return float(sum(t.numel() for t in ts if t is not None))


stft = _staged("stft", stft, lambda a, k, r: (0.0, 4.0 * _numel(a[0], r)))
istft = _staged("istft", istft, lambda a, k, r: (0.0, 4.0 * _numel(a[0], r)))
def dense_nt(x, weight, bias=None, act=ACT_NONE):
    """act(x @ weight.T + bias) for a contiguous (rows, K) matrix on the fp32 NT GEMM (CREPE's Toeplitz layers and classifier)."""
    return _linear_last_raw(x.view(1, 1, x.shape[0], x.shape[1]), weight, bias, act=act, fp32=True).view(x.shape[0], weight.shape[0])


_linear_last_raw = linear_last
_gemm_work = lambda a, k, r: (2.0 * r.numel() * a[1].shape[1], 4.0 * _numel(a[0], a[1], r, k.get("res")))
linear_last = _staged("tdf_gemm_nt", linear_last, _gemm_work)
tdf_pair = _staged("tdf_pair", tdf_pair, lambda a, k, r: (4.0 * a[0].numel() * (a[1].numel() // a[0].shape[3]),   # 2 GEMMs of R x F x H
                                                          4.0 * _numel(a[0], r, a[1], a[1])))    # x once (re-read for the residual from L2), out, weights
dense_nt = _staged("dense_gemm_nt", dense_nt, _gemm_work)
attention = _staged("attention", attention, lambda a, k, r: (4.0 * a[0].numel() * a[0].shape[1],      # 4 T^2 D per head
                                                             4.0 * _numel(a[0], a[1], a[2], r)))
sine_source = _staged("sine_source", sine_source, lambda a, k, r: (0.0, 4.0 * _numel(a[0], a[1], r)))
layernorm_ct = _staged("layernorm", layernorm_ct, lambda a, k, r: (0.0, 4.0 * _numel(a[0], k.get("res"), r)))
rownorm_act = _staged("groupnorm_gelu", rownorm_act, lambda a, k, r: (0.0, 4.0 * _numel(a[0], r)))
