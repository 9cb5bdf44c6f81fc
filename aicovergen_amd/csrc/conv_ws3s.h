// Split-precision variant of the 16-byte-fragment convolution (opt-in: AICG_PRECISION=bf16x3).
//
// Every fp32 operand is carried as two bf16 numbers, x = hi + lo (hi = bf16(x) round-to-nearest-even, lo = bf16(x - hi): 16
// significand bits survive), and every product as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// The dropped lo*lo term is <= 2^-16 relative to the product; measured on the layers of the hot path the result differs from the
// fp32-MFMA kernels by ~1e-5 relative RMS (tests/test_conv_split.py states the tolerance).  One K16 step of a 32 x 32 tile costs
// 3 x 32 matrix-pipe cycles against 8 x 64 on the fp32 MFMA: 5.3x less pipe time; the consumer loop then runs close to the LDS
// read rate (one ds_read_b128 per MFMA), tools/probes/bf16x3_probe.hip: 369 cycles per K16 step and wave against 1 630.
// The reference's own GPU path computes in fp16 (src/rvc.py:103-104,137-138); this is the more accurate of the two reduced forms.
//
// Layouts: K in groups of 16 channels; lane half h owns channels 8 h .. 8 h + 7 of a group (what the bf16 MFMA contracts per lane).
//   weights (third packed image in HBM, same bytes as the fp32 image)   [tap][Cin_pad/16][hi|lo][h][Mpad][8 bf16]
//   LDS weight stage                                                     [tap-in-stage][group][hi|lo][h][BM] x 16 B
//   LDS patch                                                            [group][hi|lo][h][position] x 16 B   (a tap = a shift)
// The weight producer is conv_ws3's (a stage is a run of 16-byte slabs in the same order as the image); the patch producer loads
// the 8 channels of an item with 8 coalesced dword loads, applies the fused pre-activation, splits, and writes two ds_write_b128.
#pragma once
#include "conv_ws3.h"

namespace aicg {

template <int BM, int XI, int KS, bool BOOST>
__device__ __forceinline__ void ws3s_produce(const ConvArgs& p, float* xs0, float* ws0, int ptid, int n, int g, int h0, int w0, int m_base,
                                             int nstages) {
    constexpr int PNT = 256;
    constexpr int WR = Ws3Geom<BM, KS>::WR;
    constexpr int WS_ELEMS = Ws3Geom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = 2 * XI * PNT * 4;   // an item = 8 channels = two float4 (hi, lo)
    const float* xg = p.x + (long)n * p.x_sn + (long)g * p.Cin_g * p.x_sc;
    const float* wg = p.wsplit + (long)g * p.w_group_stride + (long)m_base * 4;
    if (BOOST) __builtin_amdgcn_s_setprio(2);
    unsigned poff[XI], woff[WR];
    int pdst[XI];   // float4 index of the item's hi word in the patch buffer (lo = + 2 planes)
    const int items_per_chunk = (p.BKC >> 3) * p.CHS;   // (group, half) x positions
    {
        const int hin0 = h0 * p.sh - p.ph, win0 = w0 * p.sw - p.pw;
#pragma unroll
        for (int e = 0; e < XI; ++e) {
            const int idx = ptid + e * PNT;
            const int gh = (int)__umulhi((unsigned)idx, p.div_chs);   // = 2 * group + half
            const int rem = idx - gh * p.CHS;
            const int r = (int)__umulhi((unsigned)rem, p.div_twp);
            const int col = rem - r * p.TWp;
            const int hin = hin0 + r, win = win0 + col;
            const bool ok = idx < items_per_chunk && col < p.TW_in && hin >= 0 && hin < p.H && win >= 0 && win < p.W;
            poff[e] = ok ? 4u * (unsigned)(8 * gh * p.x_sc + hin * p.x_sh + win) : kBufOob;
            pdst[e] = idx < items_per_chunk ? (4 * (gh >> 1) + (gh & 1)) * p.CHS + rem : -1;
        }
        const int slabs_per_tap = p.BKC >> 2, slab_tap_stride = p.Cin_pad >> 2;
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = ptid + e * PNT;
            const int slab = idx4 / BM;
            const int m = idx4 - slab * BM;
            const int tt = slab / slabs_per_tap, sl = slab - tt * slabs_per_tap;
            const bool ok = m_base + m < p.Mpad && slab < KS / 4;
            woff[e] = ok ? 16u * (unsigned)((tt * slab_tap_stride + sl) * p.Mpad + m) : kBufOob;
        }
    }
    const unsigned ch1 = 4u * (unsigned)p.x_sc;   // byte distance of two channels
    const bool lrelu_max = p.pre_act == AICG_ACT_LRELU && p.pre_slope >= 0.f && p.pre_slope <= 1.f;
    auto load = [&](int c, int tap0, float4 (&wv)[WR], float (&xv)[XI][8]) {
        const long wbase = ((long)tap0 * (p.Cin_pad >> 2) + (long)c * (p.BKC >> 2)) * p.Mpad * 4;
        const BufRsrc wb = make_buf(wg + wbase, (unsigned)lmin(((long)p.taps * p.Cin_pad * p.Mpad - wbase - (long)m_base * 4) * 4, 0x7fffffffL));
#pragma unroll
        for (int e = 0; e < WR; ++e) wv[e] = buf_load_f32x4(wb, woff[e]);
        if (tap0 == 0) {
            const long left = (long)(p.Cin_g - c * p.BKC) * p.x_sc * 4;
            const BufRsrc xb = make_buf(xg + (long)c * p.BKC * p.x_sc, (unsigned)lmin(left, 0x7fffffffL));
#pragma unroll
            for (int e = 0; e < XI; ++e)
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[e][k] = buf_load_f32(xb, poff[e] + (unsigned)k * ch1);
        }
    };
    auto commit = [&](int st, int c, int tap0, float4 (&wv)[WR], float (&xv)[XI][8]) {
        if (tap0 == 0) {
            float4* xs = reinterpret_cast<float4*>(xs0 + (c & 1) * XS_ELEMS);
#pragma unroll
            for (int e = 0; e < XI; ++e) {
                unsigned hw[4], lw[4];
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    float v0 = xv[e][k], v1 = xv[e][k + 1];
                    // (0 <= slope <= 1: lrelu(v) = max(v, slope v), same bits for every finite v, one VALU op less -- a producer wave
                    //  gets about one VALU issue slot per MFMA while the consumers keep the matrix pipe busy)
                    if (lrelu_max) { v0 = fmaxf(v0, v0 * p.pre_slope); v1 = fmaxf(v1, v1 * p.pre_slope); }
                    else if (p.pre_act == AICG_ACT_LRELU) { v0 = v0 > 0.f ? v0 : v0 * p.pre_slope; v1 = v1 > 0.f ? v1 : v1 * p.pre_slope; }
                    else if (p.pre_act != AICG_ACT_NONE) { v0 = apply_act(v0, p.pre_act, p.pre_slope); v1 = apply_act(v1, p.pre_act, p.pre_slope); }
                    split_bf16_pair(v0, v1, hw[k >> 1], lw[k >> 1]);
                }
                if (pdst[e] >= 0) {
                    xs[pdst[e]] = make_float4(__builtin_bit_cast(float, hw[0]), __builtin_bit_cast(float, hw[1]), __builtin_bit_cast(float, hw[2]),
                                              __builtin_bit_cast(float, hw[3]));
                    xs[pdst[e] + 2 * p.CHS] = make_float4(__builtin_bit_cast(float, lw[0]), __builtin_bit_cast(float, lw[1]),
                                                          __builtin_bit_cast(float, lw[2]), __builtin_bit_cast(float, lw[3]));
                }
            }
        }
        float* ws = ws0 + (st & 1) * WS_ELEMS + ptid * 4;
#pragma unroll
        for (int e = 0; e < WR; ++e) *reinterpret_cast<float4*>(ws + e * PNT * 4) = wv[e];
    };
    auto next = [&](int& c, int& tap0) {
        tap0 += p.TT;
        if (tap0 >= p.taps) { tap0 = 0; ++c; }
    };
    float4 wv[WR];
    float xv[XI][8];
    int c = 0, t = 0;
    load(c, t, wv, xv);
    for (int st = 0; st < nstages; ++st) {
        commit(st, c, t, wv, xv);
        next(c, t);
        if (st + 1 < nstages) load(c, t, wv, xv);
        lds_barrier();
    }
}

template <int BM, int BN, int WM, int WN, int XI, int KS, bool GEN>
__global__ void __launch_bounds__(64 * (WM * WN + 4), (WM * WN == 4 ? 4 : 3)) conv_ws3s_kernel(ConvArgs p) {
    constexpr int CW = WM * WN, CNT = 64 * CW, PNT = 256;
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    constexpr int WS_ELEMS = Ws3Geom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = 2 * XI * PNT * 4;
    constexpr bool DB = TM * TN <= 3;   // two fragment register sets where 128 registers allow it
    HIP_DYNAMIC_SHARED(float, smem)
    float* const xs0 = smem;
    float* const ws0 = smem + 2 * XS_ELEMS;

    const int tid = threadIdx.x;
    const int bx = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int tw_i = bx % p.tiles_w;
    const int th_i = (bx / p.tiles_w) % p.tiles_h;
    const int n = bx / (p.tiles_w * p.tiles_h);
    const int w0 = tw_i * p.TW, h0 = th_i * p.TH;
    const int m_base = blockIdx.y * BM;
    const int g = blockIdx.z;
    const int stages_per_chunk = (p.taps + p.TT - 1) / p.TT;
    const int nstages = p.nchunk * stages_per_chunk;
    if (tid >= CNT) {
        ws3s_produce<BM, XI, KS, CW == 8>(p, xs0, ws0, tid - CNT, n, g, h0, w0, m_base, nstages);
        return;
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (TN * 32) + j * 32 + l31;
        const int jh = nl >> p.TWlog2, jw = nl & (p.TW - 1);
        boff[j] = jh * p.sh * p.TWp + jw * p.sw + half * p.CHS;
    }
#ifdef AICG_CONV_TRACE
    const int trace_wg = (wave == 0 && blockIdx.y == 0 && blockIdx.z == 0) ? (int)blockIdx.x : (1 << 30);
    trace_mark(trace_wg, 0);
    trace_val(trace_wg, 5, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));   // HW_REG_HW_ID
#else
    const int trace_wg = 0;
#endif
    trace_mark(trace_wg, 6);
    f32x16 acc[TM][TN];
    ws_init_acc32<TM, TN>(p, acc, g, m_base + wm * (TM * 32), half);
    const int a_off = wm * (TM * 32) + l31 + half * BM;
#ifdef AICG_CONV_TRACE
    if (acc[0][0][0] == 1.2345e-30f) return;
#endif
    trace_mark(trace_wg, 7);
    {
        int c = 0, tap0 = 0;
        const int gpt = p.BKC >> 4;   // 16-channel groups per tap
        for (int st = 0; st < nstages; ++st) {
            trace2(trace_wg, 0, st, 0);
            lds_barrier();
            trace2(trace_wg, 0, st, 1);
            if (st == 0) trace_mark(trace_wg, 1);
            const float4* xs = reinterpret_cast<const float4*>(xs0 + (c & 1) * XS_ELEMS);
            const float4* wt = reinterpret_cast<const float4*>(ws0 + (st & 1) * WS_ELEMS) + a_off;
            const int nt = imin(p.TT, p.taps - tap0);
            const int ngroups = nt * gpt;
            const int kh0 = tap0 / p.KW;
            int kw = tap0 - kh0 * p.KW, gg = 0;
            int xoff = kh0 * p.dh * p.TWp + kw * p.dw;
            const int step_g = 4 * p.CHS, lo_b = 2 * p.CHS, next_tap = p.dw - gpt * 4 * p.CHS, next_row = p.dh * p.TWp - p.KW * p.dw;
            struct Frag { float4 ah[TM], al[TM], bh[TN], bl[TN]; };
            auto fetch = [&](Frag& f, int s) {
                const float4* xt = xs + xoff;
#pragma unroll
                for (int i = 0; i < TM; ++i) { f.ah[i] = wt[s * 4 * BM + i * 32]; f.al[i] = wt[s * 4 * BM + 2 * BM + i * 32]; }
#pragma unroll
                for (int j = 0; j < TN; ++j) { f.bh[j] = xt[boff[j]]; f.bl[j] = xt[boff[j] + lo_b]; }
                xoff += step_g;
                if (++gg == gpt) { gg = 0; xoff += next_tap; if (++kw == p.KW) { kw = 0; xoff += next_row; } }
            };
            auto mma = [&](Frag& f) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {   // small terms first
                        acc[i][j] = mfma_bf16_32x32x16(f.al[i], f.bh[j], acc[i][j]);
                        acc[i][j] = mfma_bf16_32x32x16(f.ah[i], f.bl[j], acc[i][j]);
                        acc[i][j] = mfma_bf16_32x32x16(f.ah[i], f.bh[j], acc[i][j]);
                    }
            };
            if (DB) {
                Frag f0, f1;
                fetch(f0, 0);
                int s = 0;
                for (; s + 2 <= ngroups; s += 2) {
                    fetch(f1, s + 1);
                    mma(f0);
                    if (s + 2 < ngroups) fetch(f0, s + 2);
                    mma(f1);
                }
                if (s < ngroups) mma(f0);
            } else {
                // (measured: reloading the A fragments row by row behind their MFMAs with the B fragments a k-step ahead -- no k-step
                //  waiting for its own reads -- changes nothing: the second workgroup's wave on the SIMD already fills those gaps,
                //  and the split kernels run at the chip's power limit)
                Frag f0;
                for (int s = 0; s < ngroups; ++s) {
                    fetch(f0, s);
                    mma(f0);
                }
            }
            tap0 += p.TT;
            if (tap0 >= p.taps) { tap0 = 0; ++c; }
        }
    }
    trace_mark(trace_wg, 2);
    const bool interior = m_base + BM <= p.Cout_g && h0 + p.TH <= p.Ho && w0 + p.TW <= p.Wo;
    if (!GEN && interior && p.wide_ok) {
        lds_barrier();
        ws_epilogue32_wide<TM, TN>(p, acc, n, g, m_base + wm * (TM * 32), wn * (TN * 32), h0, w0, lane, smem + wave * kEpiScratch);
    } else {
        ws_epilogue32<TM, TN, GEN>(p, acc, n, g, m_base + wm * (TM * 32), wn * (TN * 32), h0, w0, l31, half, interior);
    }
    trace_mark(trace_wg, 3);
    trace_val(trace_wg, 4, (unsigned long long)nstages);
}

// returns 0 launched, < 0 error, 1 the configuration does not fit this form
template <int BM, int BN, int WM, int WN, int KS>
static int launch_conv_ws3s(ConvArgs& p, hipStream_t stream) {
    if (p.Cin_g < 16 || !p.wsplit) return 1;
    p.TW = choose_tile_width(p, BN);
    p.TWlog2 = ilog2(p.TW);
    p.TH = BN / p.TW;
    p.TH_in = (p.TH - 1) * p.sh + (p.KH - 1) * p.dh + 1;
    p.TW_in = (p.TW - 1) * p.sw + (p.KW - 1) * p.dw + 1;
    p.TWp = p.TW_in | 1;
    p.CHS = p.TH_in * p.TWp;
    p.tiles_w = idiv_up(p.Wo, p.TW);
    p.tiles_h = idiv_up(p.Ho, p.TH);
    p.BKC = 32;
    while (p.BKC > 16 && (p.BKC * p.CHS > 16 * 256 || p.BKC >= 2 * p.Cin_g)) p.BKC >>= 1;
    p.BKClog2 = ilog2(p.BKC);
    {
        const int cap = imax(1, KS / p.BKC);
        const int nstg = idiv_up(p.taps, cap);
        p.TT = idiv_up(p.taps, nstg);
    }
    p.nchunk = idiv_up(p.Cin_g, p.BKC);
    p.xs_total = p.BKC * p.CHS;
    p.xs_elems = (p.xs_total + 3) & ~3;
    p.div_chs = div_mul(p.CHS);
    p.div_twp = div_mul(p.TWp);
    if (p.xs_total > 24 * 256) return 1;   // <= 3 items of 8 channels per producer thread (3 only for 16-channel chunks of wide patches)
    const int xi = imax(1, idiv_up(p.xs_total / 8, 256));
    const size_t lds = (size_t)(2 * 2 * xi * 256 * 4 + 2 * Ws3Geom<BM, KS>::WS_ELEMS) * sizeof(float);
    const bool off_ok = (long)(p.BKC + 8) * p.x_sc + (long)p.H * p.x_sh < (1L << 29) && (long)p.taps * p.Cin_pad * p.Mpad < (1L << 29);
    if (lds > 160 * 1024 || !off_ok || (long)p.xs_total * p.CHS >= (1L << 32)) return 1;
    const long gx = (long)p.N * p.tiles_h * p.tiles_w;
    if (gx > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    dim3 grid((unsigned)gx, (unsigned)idiv_up(p.Cout_g, BM), (unsigned)p.groups);
    dim3 block(64 * (WM * WN + 4));
    const bool gen = p.shuffle || p.res_mul;
    p.stagger = p.stagger_first = 0;
    {
        AICG_SWITCH(wide, "AICG_CONV_WIDE", 1);
        p.wide_ok = wide && (size_t)(WM * WN) * kEpiScratch * sizeof(float) <= lds ? conv_wide_ok(p) : 0;
    }
    auto kern = gen ? (xi == 1 ? conv_ws3s_kernel<BM, BN, WM, WN, 1, KS, true>
                       : xi == 2 ? conv_ws3s_kernel<BM, BN, WM, WN, 2, KS, true> : conv_ws3s_kernel<BM, BN, WM, WN, 3, KS, true>)
                    : (xi == 1 ? conv_ws3s_kernel<BM, BN, WM, WN, 1, KS, false>
                       : xi == 2 ? conv_ws3s_kernel<BM, BN, WM, WN, 2, KS, false> : conv_ws3s_kernel<BM, BN, WM, WN, 3, KS, false>);
    allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return check_launch("conv_ws3s_kernel");
}

// instantiation units (conv_ws3s_*.hip)
int run_ws3s_128x128(ConvArgs& p, hipStream_t st);
int run_ws3s_96x128(ConvArgs& p, hipStream_t st);
int run_ws3s_64x128(ConvArgs& p, hipStream_t st);
int run_ws3s_64x64(ConvArgs& p, hipStream_t st);
int run_ws3s_32x256(ConvArgs& p, hipStream_t st);
int run_ws3s_32x128(ConvArgs& p, hipStream_t st);
int run_ws3s_64x256(ConvArgs& p, hipStream_t st);      // 4 consumers x (64 x 64): 12 MFMAs per 8 fragment reads

}  // namespace aicg
