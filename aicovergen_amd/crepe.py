"""CREPE f0 estimator on the gfx950 kernels: the slice of torchcrepe 0.0.20 that the reference uses for
f0_method='mangio-crepe' / 'mangio-crepe-tiny' (reference src/vc_infer_pipeline.py:96-137, 314-321):
torchcrepe.predict(audio, 16000, hop, 50, 1100, model, batch_size=2*hop, pad=True) -> NaN gating -> np.interp to p_len.

Frames are normalised by a reduction kernel, the first three k x 1 convolutions run through the implicit-GEMM MFMA kernel
(the k=512 / stride-4 first layer as a 4-phase k=128 convolution), BatchNorm (applied after the ReLU in CREPE) is fused
with the 2:1 max-pool, and the per-batch softmax + Viterbi decode (360 states, float64, first-index argmax like numpy) runs
one workgroup per 2*hop-frame batch.

The last three convolutions see 32 / 16 / 8 input positions per frame under a 64-tap "same"-padded kernel: 50 ... 87 % of every
output's taps multiply padding zeros, and a convolution tile (128 positions of ONE frame) would be 75 ... 94 % empty (measured r3:
9 TFLOP/s on the 256 -> 512 layer, 457 ms of a 2.25 s C4 step).  They are contracted only over the real inputs instead:
out[n][(co, wo)] = sum_{ci, wi} Wt[(co, wo)][(ci, wi)] x[n][(ci, wi)] with the Toeplitz-expanded weight
Wt[(co, wo)][(ci, wi)] = w[co][ci][wi - wo + 31] (zero outside the kernel), one NT GEMM per layer over the frames of a batch
(aicg_gemm_nt: both operands contiguous along the contraction; bias + ReLU in its epilogue).  Same products, the exact zeros of
the padding skipped; the classifier (a k = 4 kernel on 4 positions, i.e. already dense) takes the same GEMM with a sigmoid."""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops

WINDOW, PITCH_BINS = 1024, 360
BN_EPS = 0.0010000000474974513
DITHER = None   # tests: callable n_frames -> cents offsets replacing torchcrepe's random triangular dither everywhere


def _freq_to_bin(freq, ceil=False):
    b = (1200.0 * np.log2(freq / 10.0) - 1997.3794084376191) / 20
    return int(np.ceil(b) if ceil else np.floor(b))


class Crepe:
    @ops.fp32_only   # pitch-bin selection stays bit-exact under AICG_PRECISION=bf16x3
    def __init__(self, state_dict, device):
        sd = state_dict
        dev = torch.device(device)
        self.device = dev
        self.layers = []
        width = 128                                             # input positions per frame of conv2 (1024 / 4 / 2)
        for i in range(6):
            n = "conv%d" % (i + 1)
            w = sd[n + ".weight"].float()[..., 0]             # (Cout, Cin, k)
            b = sd[n + ".bias"].float()
            s = sd[n + "_BN.weight"].float() / torch.sqrt(sd[n + "_BN.running_var"].float() + BN_EPS)
            t = sd[n + "_BN.bias"].float() - sd[n + "_BN.running_mean"].float() * s
            if i == 0:
                co, _, k = w.shape                              # Conv(1 -> C, k = 512, stride 4) as 4 phases x k = 128
                w4 = w.view(co, k // 4, 4).permute(0, 2, 1).contiguous()
                pc = ops.PackedConv(w4, b, device=dev)
            elif 2 * width <= w.shape[-1]:                      # at least half of every output's taps fall on the padding
                pc = _ToeplitzGemm(w, b, width, 31, dev)
                width //= 2
            elif width < 128:
                # a frame offers fewer positions than a 128-position tile: the same 64-tap convolution as a 1 x 64 kernel over the
                # (frame, position) plane -- tiles then span several frames (a view of the batch, no data movement)
                pc = _RowConv(w, b, 31, dev)
                width //= 2
            else:
                pc = ops.PackedConv(w, b, padding=31, padding_end=32, device=dev)
                width //= 2
            self.layers.append((pc, s.contiguous().to(dev), t.contiguous().to(dev)))
        wf = sd["classifier.weight"].float()                    # (360, 4*C): feature index = h*C + c
        c_last = wf.shape[1] // 4
        # x reaches the classifier as (B, C, 4): feature (c, h) at c * 4 + h
        self.fc_w = wf.view(PITCH_BINS, 4, c_last).permute(0, 2, 1).reshape(PITCH_BINS, 4 * c_last).contiguous().to(dev)
        self.fc_b = sd["classifier.bias"].float().contiguous().to(dev)

    def __call__(self, frames):
        """frames (B, 1024) normalised -> (B, 360) sigmoid posteriors."""
        b = frames.shape[0]
        xp = F.pad(frames, (254, 254))                                      # zero padding of the first conv
        x = xp.view(b, 383, 4).transpose(1, 2).contiguous()                 # 4-phase view X[ph][q] = xpad[4q + ph]
        for i, (pc, s, t) in enumerate(self.layers):
            x = pc(x) if isinstance(pc, (_ToeplitzGemm, _RowConv)) else ops.conv(x, pc, act=ops.ACT_RELU)
            x = ops.affine_maxpool2(x, s, t)
        return ops.dense_nt(x.view(b, -1), self.fc_w, self.fc_b, act=ops.ACT_SIGMOID)


class _RowConv:
    """relu(conv1d(x, w, b)) of a (B, Cin, W) batch, padded `pad_left` / k - 1 - pad_left, evaluated as a 2-D convolution with a 1 x k
    kernel on the batch re-laid-out as ONE (1, Cin, B, W) image: the implicit-GEMM tiles (128 positions) then cover 128 / W frames
    instead of leaving 1 - W / 128 of a tile empty.  (A strided view is not enough: the kernel's range-checked buffer loads assume
    a channel's rows lie inside its channel stride; the two re-layouts move 2 x 67 MB per 2048-frame batch, ~50 us.)"""

    def __init__(self, w, b, pad_left, dev):
        k = w.shape[-1]
        self.pc = ops.PackedConv(w.unsqueeze(2), b, padding=(0, pad_left), padding_end=(0, k - 1 - pad_left), device=dev)

    def __call__(self, x):
        b, c, w = x.shape
        y = ops.conv(x.permute(1, 0, 2).contiguous().unsqueeze(0), self.pc, act=ops.ACT_RELU)   # (1, Cout, B, W)
        return y[0].permute(1, 0, 2).contiguous()


class _ToeplitzGemm:
    """relu(conv1d(x, w, b)) of a (B, Cin, W) batch with `pad_left` zeros before and k - 1 - pad_left after each frame (output
    width W), for W <= k / 2, as ONE NT GEMM over the real inputs only (module docstring)."""

    def __init__(self, w, b, width, pad_left, dev):
        co, ci, k = w.shape
        wi = torch.arange(width).view(1, width)
        wo = torch.arange(width).view(width, 1)
        tap = wi - wo + pad_left                                            # (wo, wi): out[wo] += w[tap] * x[wi]
        ok = (tap >= 0) & (tap < k)
        wt = w[:, :, tap.clamp(0, k - 1)] * ok.to(w.dtype)                  # (co, ci, wo, wi)
        self.w = wt.permute(0, 2, 1, 3).reshape(co * width, ci * width).contiguous().to(dev)
        self.b = b.repeat_interleave(width).contiguous().to(dev)
        self.cout, self.width = co, width

    def __call__(self, x):
        b = x.shape[0]
        assert x.is_contiguous() and x.shape[2] == self.width
        return ops.dense_nt(x.view(b, -1), self.w, self.b, act=ops.ACT_RELU).view(b, self.cout, self.width)


def load_crepe(model, device):
    """torchcrepe ships its weights inside the pip package (assets/full.pth / tiny.pth)."""
    import importlib.util
    import os
    spec = importlib.util.find_spec("torchcrepe")
    if spec is None or not spec.submodule_search_locations:
        raise RuntimeError("CREPE weights not found: torchcrepe (which bundles assets/%s.pth) is not installed; pass a "
                           "state_dict to VC.model_crepe instead" % model)
    path = os.path.join(list(spec.submodule_search_locations)[0], "assets", "%s.pth" % model)
    return Crepe(torch.load(path, map_location="cpu"), device)


def predict(net, audio, hop, fmin=50.0, fmax=1100.0, batch_size=None, dither=None, frame_batch=2048, group=None):
    """torchcrepe.predict for a 16 kHz (N,) waveform on net.device -> (pitch float32 (n_frames,), bins int64, posteriors).
    `dither` replaces torchcrepe's random triangular dither of the bin centres (None draws it from torch's RNG).
    `group`: the frames are independent up to the Viterbi pass, so the ranks of a torch.distributed group each run the network
    on a contiguous slice of them and all-gather the 360-bin posteriors; the decode then runs identically on
    every rank."""
    from . import dist as adist
    dev = net.device
    a = torch.as_tensor(audio).float().to(dev).view(-1)
    total = 1 + a.numel() // hop
    ap = F.pad(a, (WINDOW // 2, WINDOW // 2))
    rank, world = adist.world(group)
    per = (total + world - 1) // world
    f0, f1 = min(rank * per, total), min((rank + 1) * per, total)
    mine = torch.zeros((per, PITCH_BINS), dtype=torch.float32, device=dev)
    for i in range(f0, f1, frame_batch):
        n = min(frame_batch, f1 - i)
        fr = ap[i * hop:].unfold(0, WINDOW, hop)[:n].contiguous()          # frame extraction (re-indexing)
        mine[i - f0:i - f0 + n] = net(ops.frame_normalize(fr))
    post = adist.all_gather_equal(mine, group)[:total]
    batch_size = batch_size or total
    n_seq = (total + batch_size - 1) // batch_size
    probs = torch.zeros((n_seq, PITCH_BINS, batch_size), dtype=torch.float32, device=dev)
    lens = []
    for s in range(n_seq):
        seg = post[s * batch_size:(s + 1) * batch_size]
        probs[s, :, :seg.shape[0]] = seg.t()
        lens.append(seg.shape[0])
    bins = ops.crepe_viterbi(probs, lens, _freq_to_bin(fmin), _freq_to_bin(fmax, ceil=True))
    bins = torch.cat([bins[s, :lens[s]] for s in range(n_seq)])
    cents = (20 * bins + 1997.3794084376191).float()
    if dither is None and DITHER is not None:
        dither = DITHER(bins.numel())
    if dither is None:
        u = torch.rand(2, bins.numel(), device=dev)
        dither = (u[0] + u[1] - 1.0) * 20.0                                # triangular on (-20, 20) cents
    cents = cents + torch.as_tensor(dither, dtype=torch.float32, device=dev)
    return (10 * 2 ** (cents / 1200)), bins, post


def mangio_crepe_f0(net, x, p_len, hop, dither=None, group=None):
    """VC.get_f0_crepe_computation (reference src/vc_infer_pipeline.py:96-137)."""
    x = x.astype(np.float32)
    x = x / np.quantile(np.abs(x), 0.999)
    pitch, bins, post = predict(net, x, hop, 50.0, 1100.0, batch_size=hop * 2, dither=dither, group=group)
    p_len = p_len or x.shape[0] // hop
    source = pitch.cpu().float().numpy()
    source[source < 0.001] = np.nan
    target = np.interp(np.arange(0, len(source) * p_len, len(source)) / p_len, np.arange(0, len(source)), source)
    return np.nan_to_num(target)


def official_crepe_f0(net, x, hop, fmin=50.0, fmax=1100.0, dither=None, group=None):
    """VC.get_f0_official_crepe_computation (reference src/vc_infer_pipeline.py:139-165):
    f0, pd = torchcrepe.predict(audio, 16000, hop, fmin, fmax, model, batch_size=512, return_periodicity=True);
    pd = torchcrepe.filter.median(pd, 3); f0 = torchcrepe.filter.mean(f0, 3); f0[pd < 0.1] = 0.
    The periodicity is the posterior at the Viterbi-decoded bin (torchcrepe.core.periodicity); both 3-frame filters run in
    ops.filter3 (wave-shuffle neighbours)."""
    x = np.asarray(x, dtype=np.float32)
    pitch, bins, post = predict(net, x, hop, fmin, fmax, batch_size=512, dither=dither, group=group)
    pd = post.gather(1, bins.view(-1, 1)).view(-1)
    pd = ops.filter3(pd, "median")
    f0 = ops.filter3(pitch, "mean")
    f0 = torch.where(pd < 0.1, torch.zeros_like(f0), f0)
    return f0.cpu().numpy()
