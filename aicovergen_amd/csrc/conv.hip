// Implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered
// fmaf chain), one kernel family for every dense contraction on the hot path:
//   * NSF-HiFiGAN ResBlock1 dilated Conv1d + fused leaky-ReLU prologue / residual epilogue
//       (reference src/infer_pack/modules.py:299-312, models.py:494-516)
//   * ConvTranspose1d/2d as a 1x1 GEMM (followed by aicg_col2im)        (models.py:453-463, rmvpe.py:147-155)
//   * WaveNet / flow / FFN / 1x1 projections of enc_p                   (modules.py:188-213, attentions.py:391-399)
//   * HuBERT feature-extractor strided convs, grouped positional conv, QKV / FFN linears
//       (fairseq HubertModel as called at src/vc_infer_pipeline.py:398-406)
//   * RMVPE and MDX-Net 3x3 Conv2d + folded BatchNorm + ReLU (+ residual) (rmvpe.py:23-58, mdx.py:74-77)
//
// GEMM view per group:  M = Cout_g,  N = Ho*Wo (tiled as TH x TW output patches, TW a power of two),
// K = Cin_g*KH*KW ordered [channel chunk][tap][channel in chunk].
// A (weights) is pre-packed on the host as [K][Mpad] so a stage is one coalesced float4 copy into LDS;
// B is never materialised: a chunk of BKC input channels of the input patch (with halo) is staged
// into LDS once -- with the fused pre-activation applied once per element, not once per tap -- and
// every tap reads it at a shifted offset.  Lane l of a wave feeds the MFMA with
// A[k = l>>5][m = l&31] and B[k = l>>5][n = l&31]: both are unit-stride, conflict-free ds_read_b32.
#include "common.h"

namespace aicg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* res;
    float* y;
    int N, Cin_g, H, W, Cout_g, Ho, Wo, KH, KW, sh, sw, ph, pw, dh, dw, groups;
    long x_sn, x_sc, x_sh, y_sn, y_sc, y_sh, r_sn, r_sc, r_sh;
    int pre_act;
    float pre_slope;
    int act;
    float act_slope;
    float out_scale;
    int accumulate;
    int res_first;
    // derived tiling
    int TW, TWlog2, TH, TH_in, TW_in, TWp, CHS, BKC, TT, tiles_w, tiles_h, nchunk, taps, Mpad, xs_elems;
    long w_group_stride;
};

static constexpr int KSTAGE = 32;  // K rows of packed weights staged per barrier pair

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs p) {
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    static_assert(WM * WN == 4, "4 waves per workgroup");
    HIP_DYNAMIC_SHARED(float, smem)
    float* xs = smem;
    float* ws = smem + p.xs_elems;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    const int bx = blockIdx.x;
    const int tw_i = bx % p.tiles_w;
    const int th_i = (bx / p.tiles_w) % p.tiles_h;
    const int n = bx / (p.tiles_w * p.tiles_h);
    const int w0 = tw_i * p.TW, h0 = th_i * p.TH;
    const int m_base = blockIdx.y * BM;
    const int g = blockIdx.z;

    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (TN * 32) + j * 32 + l31;
        const int jh = nl >> p.TWlog2, jw = nl & (p.TW - 1);
        boff[j] = jh * p.sh * p.TWp + jw * p.sw + half * p.CHS;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* xg = p.x + (long)n * p.x_sn + (long)g * p.Cin_g * p.x_sc;
    const float* wg = p.w + (long)g * p.w_group_stride;
    const int hin0 = h0 * p.sh - p.ph, win0 = w0 * p.sw - p.pw;
    const int a_off = wm * (TM * 32) + l31 + half * BM;
    const int xs_total = p.BKC * p.CHS;

    for (int c = 0; c < p.nchunk; ++c) {
        __syncthreads();  // every wave is done reading the previous chunk's tiles
        // ---- stage BKC channels of the input patch (halo included), pre-activation fused ----------
        for (int idx = tid; idx < xs_total; idx += 256) {
            const int ci = idx / p.CHS;
            const int rem = idx - ci * p.CHS;
            const int r = rem / p.TWp;
            const int col = rem - r * p.TWp;
            const int cg = c * p.BKC + ci;
            const int hin = hin0 + r, win = win0 + col;
            float v = 0.f;
            if (cg < p.Cin_g && col < p.TW_in && hin >= 0 && hin < p.H && win >= 0 && win < p.W) {
                v = xg[(long)cg * p.x_sc + (long)hin * p.x_sh + win];
                v = apply_act(v, p.pre_act, p.pre_slope);
            }
            xs[idx] = v;
        }
        int kh = 0, kw = 0;
        for (int tap0 = 0; tap0 < p.taps; tap0 += p.TT) {
            const int nt = imin(p.TT, p.taps - tap0);
            const int rows = nt * p.BKC;
            if (tap0 > 0) __syncthreads();  // previous weight stage fully consumed
            // ---- stage `rows` packed weight rows x BM columns (float4, coalesced) ---------------------
            const float* wrow0 = wg + ((long)c * p.taps + tap0) * p.BKC * p.Mpad;
            for (int idx4 = tid; idx4 < rows * (BM / 4); idx4 += 256) {
                const int r = idx4 / (BM / 4);
                const int c4 = idx4 - r * (BM / 4);
                const int mcol = m_base + c4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mcol < p.Mpad) v = *reinterpret_cast<const float4*>(wrow0 + (long)r * p.Mpad + mcol);
                *reinterpret_cast<float4*>(ws + r * BM + c4 * 4) = v;
            }
            __syncthreads();
            // ---- MFMA over the staged K rows ----------------------------------------------------------
            for (int tt = 0; tt < nt; ++tt) {
                const int tapoff = kh * p.dh * p.TWp + kw * p.dw;
                const float* wt = ws + tt * p.BKC * BM + a_off;
                const float* xt = xs + tapoff;
                for (int kk = 0; kk < p.BKC; kk += 2) {
                    float a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = wt[kk * BM + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = xt[kk * p.CHS + boff[j]];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
                if (++kw == p.KW) { kw = 0; ++kh; }
            }
        }
    }

    // ---- epilogue: y = [y +] out_scale * (act(acc + bias [+ res]) [+ res]) ------------------------------------
    const long y_base = (long)n * p.y_sn, r_base = (long)n * p.r_sn;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (TN * 32) + j * 32 + l31;
        const int ho = h0 + (nl >> p.TWlog2), wo = w0 + (nl & (p.TW - 1));
        if (ho >= p.Ho || wo >= p.Wo) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= p.Cout_g) continue;
                const int co = g * p.Cout_g + m;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[co];
                float rv = 0.f;
                if (p.res) rv = p.res[r_base + (long)co * p.r_sc + (long)ho * p.r_sh + wo];
                if (p.res_first) v += rv;
                v = apply_act(v, p.act, p.act_slope);
                if (!p.res_first) v += rv;
                v *= p.out_scale;
                float* yp = p.y + y_base + (long)co * p.y_sc + (long)ho * p.y_sh + wo;
                if (p.accumulate) v += *yp;
                *yp = v;
            }
        }
    }
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <int BM, int BN, int WM, int WN>
static int launch_conv(ConvArgs& p, hipStream_t stream) {
    // output patch: TW (power of two) columns x TH rows = BN positions
    int TW = 1 << ilog2(p.Wo);
    if (TW > BN) TW = BN;
    if (p.Ho == 1) TW = BN;
    p.TW = TW;
    p.TWlog2 = ilog2(TW);
    p.TH = BN / TW;
    p.TH_in = (p.TH - 1) * p.sh + (p.KH - 1) * p.dh + 1;
    p.TW_in = (p.TW - 1) * p.sw + (p.KW - 1) * p.dw + 1;
    p.TWp = p.TW_in | 1;
    p.CHS = p.TH_in * p.TWp;
    p.tiles_w = idiv_up(p.Wo, p.TW);
    p.tiles_h = idiv_up(p.Ho, p.TH);
    p.xs_elems = (p.BKC * p.CHS + 3) & ~3;
    const size_t lds = (size_t)(p.xs_elems + KSTAGE * BM) * sizeof(float);
    if (lds > 160 * 1024)
        return fail(AICG_E_LDS, "conv: input patch of %d x %d x %d floats does not fit LDS (use aicg_conv1d_cin1 for Cin=1)",
                    p.BKC, p.TH_in, p.TWp);
    auto kern = conv_mfma_kernel<BM, BN, WM, WN>;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long gx = (long)p.N * p.tiles_h * p.tiles_w;
    if (gx > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    dim3 grid((unsigned)gx, (unsigned)idiv_up(p.Cout_g, BM), (unsigned)p.groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return check_launch("conv_mfma_kernel");
}

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_conv_bkc(int taps) {
    // channels per K chunk: keep a weight stage at KSTAGE = 32 rows (TT taps x BKC channels)
    if (taps <= 1) return 32;
    if (taps <= 3) return 16;
    return 8;
}

extern "C" int aicg_conv_desc_size(void) { return (int)sizeof(aicg_conv_desc); }

extern "C" int aicg_conv_forward(const aicg_conv_desc* d, const float* x, const float* w_packed, const float* bias,
                                 const float* res, float* y, void* stream) {
    if (!d || !x || !w_packed || !y) return fail(AICG_E_ARG, "aicg_conv_forward: null pointer");
    if (d->groups < 1 || d->Cin % d->groups || d->Cout % d->groups)
        return fail(AICG_E_SHAPE, "aicg_conv_forward: channels not divisible by groups");
    if (d->KH < 1 || d->KW < 1 || d->stride_h < 1 || d->stride_w < 1 || d->dil_h < 1 || d->dil_w < 1)
        return fail(AICG_E_SHAPE, "aicg_conv_forward: bad kernel geometry");
    int Ho = (d->H + 2 * d->pad_h - d->dil_h * (d->KH - 1) - 1) / d->stride_h + 1;
    int Wo = (d->W + 2 * d->pad_w - d->dil_w * (d->KW - 1) - 1) / d->stride_w + 1;
    // the caller may ask for fewer outputs than the geometry yields (fairseq SamePad drops the last frame)
    if (d->Ho > Ho || d->Wo > Wo || d->Ho < 0 || d->Wo < 0)
        return fail(AICG_E_SHAPE, "aicg_conv_forward: output %dx%d exceeds geometry (%dx%d)", d->Ho, d->Wo, Ho, Wo);
    Ho = d->Ho; Wo = d->Wo;
    if (d->N == 0 || Ho <= 0 || Wo <= 0) return AICG_OK;
    ConvArgs p;
    p.x = x; p.w = w_packed; p.bias = bias; p.res = res; p.y = y;
    p.N = d->N; p.Cin_g = d->Cin / d->groups; p.H = d->H; p.W = d->W; p.Cout_g = d->Cout / d->groups;
    p.Ho = Ho; p.Wo = Wo; p.KH = d->KH; p.KW = d->KW; p.sh = d->stride_h; p.sw = d->stride_w;
    p.ph = d->pad_h; p.pw = d->pad_w; p.dh = d->dil_h; p.dw = d->dil_w; p.groups = d->groups;
    p.x_sn = d->x_sn; p.x_sc = d->x_sc; p.x_sh = d->x_sh;
    p.y_sn = d->y_sn; p.y_sc = d->y_sc; p.y_sh = d->y_sh;
    p.r_sn = d->r_sn; p.r_sc = d->r_sc; p.r_sh = d->r_sh;
    p.pre_act = d->pre_act; p.pre_slope = d->pre_slope; p.act = d->act; p.act_slope = d->act_slope;
    p.out_scale = d->out_scale; p.accumulate = d->accumulate; p.res_first = d->res_before_act;
    p.taps = p.KH * p.KW;
    p.BKC = aicg_conv_bkc(p.taps);
    p.TT = KSTAGE / p.BKC;
    p.nchunk = idiv_up(p.Cin_g, p.BKC);
    p.Mpad = idiv_up(p.Cout_g, 32) * 32;
    p.w_group_stride = (long)p.nchunk * p.taps * p.BKC * p.Mpad;

    // tile shape: smallest padded M, then enough workgroups to fill 256 CUs
    const int M = p.Cout_g;
    int BM = 128;
    {
        int best = idiv_up(M, 128) * 128;
        if (idiv_up(M, 64) * 64 < best) { best = idiv_up(M, 64) * 64; BM = 64; }
        if (idiv_up(M, 32) * 32 < best) { best = idiv_up(M, 32) * 32; BM = 32; }
    }
    const long npos = (long)p.N * Ho * Wo;
    const long mt = idiv_up(M, BM) * (long)p.groups;
    hipStream_t st = (hipStream_t)stream;
    if (BM == 128) return launch_conv<128, 128, 2, 2>(p, st);
    if (BM == 64) {
        if (mt * ldiv_up(npos, 128) < 512) return launch_conv<64, 64, 2, 2>(p, st);
        return launch_conv<64, 128, 2, 2>(p, st);
    }
    if (mt * ldiv_up(npos, 256) < 512) return launch_conv<32, 128, 1, 4>(p, st);
    return launch_conv<32, 256, 1, 4>(p, st);
}
