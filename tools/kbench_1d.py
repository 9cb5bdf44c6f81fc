"""1-D conv micro-benchmark (vocoder ResBlock / HuBERT / enc_p shapes); A/B through the AICG_CONV_* env switches."""
import os, sys, torch
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from aicovergen_amd import ops  # noqa: E402
dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def case(name, ci, co, k, d, T, groups=1, stride=1, fused=True):
    x = torch.randn(1, ci, T, device=dev)
    w = torch.randn(co, ci // groups, k) * 0.05
    pc = ops.PackedConv(w, torch.randn(co), padding=(k - 1) * d // 2, dilation=d, stride=stride, groups=groups, device=dev)
    out = torch.empty(1, co, pc.out_hw(1, T)[1], device=dev)
    kw = dict(pre_act=ops.ACT_LRELU, pre_slope=0.1) if fused else {}
    res_t = x if (fused and ci == co and stride == 1) else None
    t = timeit(lambda: ops.conv(x, pc, res=res_t, out=out, **kw))
    print(f"{name:24s} {t*1e3:8.3f} ms {2.0*co*(ci//groups)*k*out.shape[-1]/t/1e12:7.1f} TF", flush=True)


print({k: v for k, v in os.environ.items() if k.startswith("AICG_")})
for (c, L) in [(256, 66000), (128, 660000), (64, 1320000), (32, 2640000)]:
    for k, d in ((3, 1), (7, 3), (11, 5)):
        case(f"rb_c{c}_k{k}_d{d}", c, c, k, d, L)
case("lin_768_3072_T3300", 768, 3072, 1, 1, 3300, fused=False)
case("lin_3072_768_T3300", 3072, 768, 1, 1, 3300, fused=False)
case("lin_768_768_T3300", 768, 768, 1, 1, 3300, fused=False)
case("ffn_192_768_k3_T6600", 192, 768, 3, 1, 6600, fused=False)
case("ffn_768_192_k3_T6600", 768, 192, 3, 1, 6600, fused=False)
case("wn_192_384_k5_T6600", 192, 384, 5, 1, 6600, fused=False)
case("hubert_fe_512_k3_s2", 512, 512, 3, 1, 105615, stride=2, fused=False)
case("posconv_768_k128_g16", 768, 768, 128, 1, 3300, groups=16, fused=False)
