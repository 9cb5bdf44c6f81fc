"""TFC-TDF U-Net of the MDX-Net separator on the gfx950 kernels.

The reference executes this network as an opaque ONNX graph through onnxruntime (src/mdx.py:74-77,193).  Here it
is rebuilt from its parameters: 3x3 convs with eval-mode BatchNorm folded (implicit-GEMM MFMA kernel), the
time-distributed Linear blocks over the frequency axis (NT GEMM kernel with the BatchNorm/ReLU/residual epilogue),
2x2 stride-2 down/up-sampling (conv / GEMM + col2im) and multiplicative skips.  The whole net runs in the
(B, C, T, F) layout the STFT kernel produces (frequency contiguous), so the two transposes of the published
graph and every host<->device copy of the reference's per-window loop disappear.
"""
import torch

from . import ops


def _bn_affine(sd, name, eps=1e-5):
    s = sd[name + ".weight"].float() / torch.sqrt(sd[name + ".running_var"].float() + eps)
    return s, sd[name + ".bias"].float() - sd[name + ".running_mean"].float() * s


def infer_cfg(sd):
    g = sd["first_conv.0.weight"].shape[0]
    n = 0
    while "ds.%d.0.weight" % n in sd:
        n += 1
    l = 0
    while "ds_dense.0.tfc.H.%d.0.weight" % l in sd:
        l += 1
    f = sd["ds_dense.0.tdf.0.weight"].shape[1]
    bn = f // sd["ds_dense.0.tdf.0.weight"].shape[0]
    return dict(dim_c=sd["first_conv.0.weight"].shape[1], g=g, n=n, l=l, k=sd["ds_dense.0.tfc.H.0.0.weight"].shape[-1], bn=bn,
                dim_f=f)


class _TfcTdf:
    def __init__(self, sd, name, l, dev):
        self.convs = []
        for j in range(l):
            p = "%s.tfc.H.%d" % (name, j)
            s, t = _bn_affine(sd, p + ".1")
            w = sd[p + ".0.weight"].float() * s.view(-1, 1, 1, 1)
            b = sd[p + ".0.bias"].float() * s + t
            self.convs.append(ops.PackedConv(w, b, padding=w.shape[-1] // 2, device=dev))
        self.w1 = sd[name + ".tdf.0.weight"].float().contiguous().to(dev)
        self.b1 = sd[name + ".tdf.0.bias"].float().contiguous().to(dev)
        self.w2 = sd[name + ".tdf.3.weight"].float().contiguous().to(dev)
        self.b2 = sd[name + ".tdf.3.bias"].float().contiguous().to(dev)
        self.s1, self.t1 = (z.contiguous().to(dev) for z in _bn_affine(sd, name + ".tdf.1"))
        self.s2, self.t2 = (z.contiguous().to(dev) for z in _bn_affine(sd, name + ".tdf.4"))
        self.w1p = self.w2p = None    # aicg_tdf_pair's packed weight images, made on first use for the geometries the fused kernel covers

    def __call__(self, x):
        for pc in self.convs:
            x = ops.conv(x, pc, act=ops.ACT_RELU)
        f, h = self.w1.shape[1], self.w1.shape[0]
        if not ops.split_precision and x.shape[2] % 32 == 0 and ops.tdf_pair_supported(f, h, x.shape[2]):
            # x + tdf(x) in one launch: the f / bn intermediate never leaves the register file (csrc/tdf_pair.hip)
            if self.w1p is None:
                self.w1p, self.w2p = ops.pack_tdf_w1(self.w1), ops.pack_tdf_w2(self.w2)
            return ops.tdf_pair(x, self.w1p, self.b1, self.s1, self.t1, self.w2p, self.b2, self.s2, self.t2)
        t = ops.linear_last(x, self.w1, self.b1, self.s1, self.t1, act=ops.ACT_RELU)
        return ops.linear_last(t, self.w2, self.b2, self.s2, self.t2, act=ops.ACT_RELU, res=x)  # x + tdf(x)


class ConvTDFNet:
    def __init__(self, state_dict, device):
        sd = state_dict
        dev = torch.device(device)
        self.device = dev
        self.cfg = infer_cfg(sd)
        n, l = self.cfg["n"], self.cfg["l"]
        s, t = _bn_affine(sd, "first_conv.1")
        self.first = ops.PackedConv(sd["first_conv.0.weight"].float() * s.view(-1, 1, 1, 1), sd["first_conv.0.bias"].float() * s + t,
                                    device=dev)
        self.ds_dense, self.ds, self.us, self.us_dense = [], [], [], []
        for i in range(n):
            self.ds_dense.append(_TfcTdf(sd, "ds_dense.%d" % i, l, dev))
            s, t = _bn_affine(sd, "ds.%d.1" % i)
            self.ds.append(ops.PackedConv(sd["ds.%d.0.weight" % i].float() * s.view(-1, 1, 1, 1), sd["ds.%d.0.bias" % i].float() * s + t,
                                          stride=2, device=dev))
        self.mid = _TfcTdf(sd, "mid_dense", l, dev)
        for i in range(n):
            s, t = _bn_affine(sd, "us.%d.1" % i)
            self.us.append(ops.PackedConvTranspose(sd["us.%d.0.weight" % i].float() * s.view(1, -1, 1, 1),
                                                   sd["us.%d.0.bias" % i].float() * s + t, stride=2, device=dev))
            self.us_dense.append(_TfcTdf(sd, "us_dense.%d" % i, l, dev))
        self.final = ops.PackedConv(sd["final_conv.0.weight"].float(), sd["final_conv.0.bias"], device=dev)

    def forward_tf(self, x):
        """x: (B, 4, T, F) frame-major spectrogram planes -> same layout."""
        x = ops.conv(x, self.first, act=ops.ACT_RELU)
        skips = []
        for dense, down in zip(self.ds_dense, self.ds):
            x = dense(x)
            skips.append(x)
            x = ops.conv(x, down, act=ops.ACT_RELU)
        x = self.mid(x)
        for i, (up, dense) in enumerate(zip(self.us, self.us_dense)):
            x = ops.conv_transpose(x, up, act=ops.ACT_RELU, mul=skips[-i - 1])  # x = relu(bn(convT(x))) * skip, one kernel
            x = dense(x)
        return ops.conv(x, self.final)

    def __call__(self, spec):
        """spec: (B, 4, dim_f, dim_t) like the ONNX graph's 'input' -> (B, 4, dim_f, dim_t)."""
        y = self.forward_tf(spec.to(self.device).float().transpose(2, 3).contiguous())
        return y.transpose(2, 3).contiguous()
