// Wave-specialised implicit-GEMM convolution, third form: 16-byte MFMA fragments.
//
// r2 micro-probe (tools/probes/mfma_loop_probe.hip, one consumer wave per SIMD, v_mfma_f32_32x32x2_f32):
//     MFMAs on constant operands                              192 cycles per k-step (TM = 3: the issue floor)
//     + ds_read_b32 fragments, results unused                 192
//     + the MFMAs consume them (the conv_ws_kernel loop)      256   (prefetch distance 1 or 2: same)
//     + fragments of four k-steps per ds_read_b128            204
// Every hand-over of LDS-loaded registers to the matrix pipe costs the wave one bubble of ~50-64 cycles, whatever the prefetch
// distance; with one hand-over per k-step that is a quarter of a 3-MFMA k-step and a third of a 2-MFMA one.  Here both
// operands are laid out in LDS so that ONE ds_read_b128 per fragment feeds FOUR k-steps:
//
//   v_mfma_f32_32x32x2_f32, k-step s: lane (l31, half) supplies A[m = l31][k = 2 s + half] and B[k = 2 s + half][n = l31].
//   K is cut into groups of 8 rows; lane `half` needs rows {half, 2 + half, 4 + half, 6 + half} of a group for its four k-steps:
//       weights  [group][parity][m][4]      element j of the float4 = row 8 group + 2 j + parity     (m contiguous: conflict-free)
//       patch    [group][parity][pos][4]    channel 8 group + 2 j + parity at patch position pos     (a tap = a shift of pos)
//   The packed weight image in HBM has the same order ([tap][Cin_pad / 8][2][Mpad][4], appended to the classic image), so a weight
//   stage is still a few coalesced float4 copies; a producer thread gathers the 4 channels of a patch quad with 4 coalesced dword
//   loads and writes one ds_write_b128.
//
// Everything else (producer / consumer roles, stage hand-over through one LDS-only barrier, buffer-resource loads with
// range-checked zero padding, tile walk, epilogue) is conv_ws_kernel's.  Layers with fewer than 8 input channels per group stay
// on conv_ws_kernel.  The 160-row tile (5 x 16 accumulators: two sets of 16-byte fragments do not fit 128 registers) keeps ONE
// set of A fragments and reloads each row's quad right behind the four MFMAs that consumed it (+3.5 ... 6 % over the
// 4-byte-fragment kernel on the 144- / 160- / 480-channel layers); its shuffle / multiplicative-skip instantiation spills inside
// the K loop, so those layers stay on conv_ws_kernel too.
#pragma once
#include "conv_kernels.h"

namespace aicg {

template <int BM, int KS>
struct Ws3Geom {
    static constexpr int PNT = 256;
    static constexpr int WR = (KS * BM / 4 + PNT - 1) / PNT;      // float4 weight copies per producer thread per stage
    // + 8 slack rows: the consumers' last (discarded) fragment prefetch of a stage reads one k-group past it
    static constexpr int WS_ELEMS = ((KS + 8) * BM > WR * PNT * 4) ? (KS + 8) * BM : WR * PNT * 4;
};

// Producer role: 256 threads stage every K stage of one output tile.  XQ = patch quads (float4 of 4 channels) per thread.
template <int BM, int XQ, int KS, bool BOOST>
__device__ __forceinline__ void ws3_produce(const ConvArgs& p, float* xs0, float* ws0, int ptid, int n, int g, int h0, int w0, int m_base,
                                            int nstages) {
    constexpr int PNT = 256;
    constexpr int WR = Ws3Geom<BM, KS>::WR;
    constexpr int WS_ELEMS = Ws3Geom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = XQ * PNT * 4;
    const float* xg = p.x + (long)n * p.x_sn + (long)g * p.Cin_g * p.x_sc;
    const float* wg = p.w3 + (long)g * p.w_group_stride + (long)m_base * 4;
    if (BOOST) __builtin_amdgcn_s_setprio(2);
    // byte offsets relative to the chunk / stage base, kBufOob for slots that must read as zero (see ws_produce)
    unsigned poff[XQ], woff[WR];
    const int quads_per_chunk = (p.BKC >> 2) * p.CHS;   // (group, parity) planes x positions
    {
        const int hin0 = h0 * p.sh - p.ph, win0 = w0 * p.sw - p.pw;
#pragma unroll
        for (int e = 0; e < XQ; ++e) {
            const int idx = ptid + e * PNT;
            const int plane = (int)__umulhi((unsigned)idx, p.div_chs);   // = 2 * group + parity
            const int rem = idx - plane * p.CHS;
            const int r = (int)__umulhi((unsigned)rem, p.div_twp);
            const int col = rem - r * p.TWp;
            const int hin = hin0 + r, win = win0 + col;
            const bool ok = idx < quads_per_chunk && col < p.TW_in && hin >= 0 && hin < p.H && win >= 0 && win < p.W;
            const int ci = 8 * (plane >> 1) + (plane & 1);               // channel of element j = 0; element j is 2 j channels on
            poff[e] = ok ? 4u * (unsigned)(ci * p.x_sc + hin * p.x_sh + win) : kBufOob;
        }
        const int slabs_per_tap = p.BKC >> 2;            // (group, parity) slabs of BM float4 per tap of a stage
        const int slab_tap_stride = (p.Cin_pad >> 2);    // slabs per tap in the packed image
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = ptid + e * PNT;
            const int slab = idx4 / BM;                  // stage-local slab: (tap-in-stage, group, parity)
            const int m = idx4 - slab * BM;
            const int tt = slab / slabs_per_tap, sl = slab - tt * slabs_per_tap;
            const bool ok = m_base + m < p.Mpad && slab < KS / 4;
            woff[e] = ok ? 16u * (unsigned)((tt * slab_tap_stride + sl) * p.Mpad + m) : kBufOob;
        }
    }
    const unsigned ch2 = 8u * (unsigned)p.x_sc;          // byte distance of two channels: element j -> j + 1 of a quad
    auto load = [&](int c, int tap0, float4 (&wv)[WR], float4 (&xv)[XQ]) {
        // stage base: slab ((tap0 * Cin_pad / 8 + c * BKC / 8) * 2) of this group's v3 image
        const long wbase = ((long)tap0 * (p.Cin_pad >> 2) + (long)c * (p.BKC >> 2)) * p.Mpad * 4;
        const BufRsrc wb = make_buf(wg + wbase, (unsigned)lmin(((long)p.taps * p.Cin_pad * p.Mpad - wbase - (long)m_base * 4) * 4, 0x7fffffffL));
#pragma unroll
        for (int e = 0; e < WR; ++e) wv[e] = buf_load_f32x4(wb, woff[e]);
        if (tap0 == 0) {  // a new channel chunk: its input patch (halo included)
            const long left = (long)(p.Cin_g - c * p.BKC) * p.x_sc * 4;  // bytes up to the end of this group's channels
            const BufRsrc xb = make_buf(xg + (long)c * p.BKC * p.x_sc, (unsigned)lmin(left, 0x7fffffffL));
#pragma unroll
            for (int e = 0; e < XQ; ++e) {
                xv[e].x = buf_load_f32(xb, poff[e]);
                xv[e].y = buf_load_f32(xb, poff[e] + ch2);
                xv[e].z = buf_load_f32(xb, poff[e] + 2u * ch2);
                xv[e].w = buf_load_f32(xb, poff[e] + 3u * ch2);
            }
        }
    };
    auto commit = [&](int st, int c, int tap0, float4 (&wv)[WR], float4 (&xv)[XQ]) {
        if (tap0 == 0) {
            float* xs = xs0 + (c & 1) * XS_ELEMS + ptid * 4;
            if (p.pre_act == AICG_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < XQ; ++e) *reinterpret_cast<float4*>(xs + e * PNT * 4) = xv[e];
            } else if (p.pre_act == AICG_ACT_LRELU && p.pre_slope >= 0.f && p.pre_slope <= 1.f) {
                // lrelu(v) = max(v, slope v) for 0 <= slope <= 1 (same bits for every finite v): a multiply and a max instead of
                // compare + multiply + select -- producer VALU slots are scarce while the consumers keep the matrix pipe busy
                const float sl = p.pre_slope;
#pragma unroll
                for (int e = 0; e < XQ; ++e) {
                    float4 v = xv[e];
                    v.x = fmaxf(v.x, v.x * sl); v.y = fmaxf(v.y, v.y * sl);
                    v.z = fmaxf(v.z, v.z * sl); v.w = fmaxf(v.w, v.w * sl);
                    *reinterpret_cast<float4*>(xs + e * PNT * 4) = v;
                }
            } else if (p.pre_act == AICG_ACT_LRELU) {
                const float sl = p.pre_slope;
#pragma unroll
                for (int e = 0; e < XQ; ++e) {
                    float4 v = xv[e];
                    v.x = v.x > 0.f ? v.x : v.x * sl; v.y = v.y > 0.f ? v.y : v.y * sl;
                    v.z = v.z > 0.f ? v.z : v.z * sl; v.w = v.w > 0.f ? v.w : v.w * sl;
                    *reinterpret_cast<float4*>(xs + e * PNT * 4) = v;
                }
            } else {
#pragma unroll
                for (int e = 0; e < XQ; ++e) {
                    float4 v = xv[e];
                    v.x = apply_act(v.x, p.pre_act, p.pre_slope); v.y = apply_act(v.y, p.pre_act, p.pre_slope);
                    v.z = apply_act(v.z, p.pre_act, p.pre_slope); v.w = apply_act(v.w, p.pre_act, p.pre_slope);
                    *reinterpret_cast<float4*>(xs + e * PNT * 4) = v;
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
    float4 xv[XQ];
    int c = 0, t = 0;
    load(c, t, wv, xv);
    for (int st = 0; st < nstages; ++st) {
        commit(st, c, t, wv, xv);
        next(c, t);
        if (st + 1 < nstages) load(c, t, wv, xv);
        lds_barrier();  // stage st published (and the consumers are done with stage st - 1)
    }
}

template <int BM, int BN, int WM, int WN, int XQ, int KS, bool GEN>
__global__ void __launch_bounds__(64 * (WM * WN + 4), (WM * WN == 4 ? 4 : 3)) conv_ws3_kernel(ConvArgs p) {
    constexpr int CW = WM * WN, CNT = 64 * CW, PNT = 256;
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    constexpr int WS_ELEMS = Ws3Geom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = XQ * PNT * 4;
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
        ws3_produce<BM, XQ, KS, CW == 8>(p, xs0, ws0, tid - CNT, n, g, h0, w0, m_base, nstages);
        return;
    }

    // ================= consumers =================
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    // float4 index of this lane's B fragment inside a (group, parity) plane pair, per column tile
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
    if (CW == 4 && p.stagger > 0) {   // see ConvArgs::stagger
        const unsigned flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (flat < (unsigned)p.stagger_first) {
            // spread the initial fill over one tile time (all CUs otherwise run their prologue loads / epilogue stores as one burst)
            const unsigned long long delay = (unsigned long long)p.stagger * ((flat * 37u) & 63u) / 64u;
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < delay) __builtin_amdgcn_s_sleep(64);
        }
    }
    trace_mark(trace_wg, 6);
    f32x16 acc[TM][TN];
    ws_init_acc32<TM, TN>(p, acc, g, m_base + wm * (TM * 32), half);
    const int a_off = wm * (TM * 32) + l31 + half * BM;   // float4 index inside a slab pair
#ifdef AICG_CONV_TRACE
    if (acc[0][0][0] == 1.2345e-30f) return;
#endif
    trace_mark(trace_wg, 7);
    {
        int c = 0, tap0 = 0;
        const int gpt = p.BKC >> 3;   // k-groups per tap
        for (int st = 0; st < nstages; ++st) {
            trace2(trace_wg, 0, st, 0);
            lds_barrier();  // stage st is in LDS
            trace2(trace_wg, 0, st, 1);
            if (st == 0) trace_mark(trace_wg, 1);
            const float4* xs = reinterpret_cast<const float4*>(xs0 + (c & 1) * XS_ELEMS);
            const float4* wt = reinterpret_cast<const float4*>(ws0 + (st & 1) * WS_ELEMS) + a_off;
            const int nt = imin(p.TT, p.taps - tap0);
            const int ngroups = nt * gpt;   // k-groups (8 rows = 4 k-steps) of this stage
            // patch float4 offset of the (tap, group) being fetched, advanced incrementally: + 2 planes per group, then to the next
            // column tap, then to the next kernel row
            const int kh0 = tap0 / p.KW;
            int kw = tap0 - kh0 * p.KW, gg = 0;
            int xoff = kh0 * p.dh * p.TWp + kw * p.dw;
            const int step_g = 2 * p.CHS, next_tap = p.dw - gpt * 2 * p.CHS, next_row = p.dh * p.TWp - p.KW * p.dw;
            float4 a0[TM], b0[TN], a1[TM], b1[TN];
            auto fetch = [&](float4 (&a)[TM], float4 (&b)[TN], int s) {
                const float4* xt = xs + xoff;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wt[s * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = xt[boff[j]];
                xoff += step_g;
                if (++gg == gpt) { gg = 0; xoff += next_tap; if (++kw == p.KW) { kw = 0; xoff += next_row; } }
            };
            auto mma = [&](float4 (&a)[TM], float4 (&b)[TN]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = u == 0 ? a[i].x : u == 1 ? a[i].y : u == 2 ? a[i].z : a[i].w;
                            const float bv = u == 0 ? b[j].x : u == 1 ? b[j].y : u == 2 ? b[j].z : b[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                        }
            };
            if constexpr (TM * TN <= 4) {
                fetch(a0, b0, 0);
                int s = 0;
                for (; s + 2 <= ngroups; s += 2) {
                    fetch(a1, b1, s + 1);
                    mma(a0, b0);
                    fetch(a0, b0, s + 2);  // unconditional: past the last group this reads (never uses) the LDS slack rows
                    mma(a1, b1);
                }
                if (s < ngroups) mma(a0, b0);
            } else {
                // 5 accumulator tiles (the 160-row tile) leave room for ONE set of A fragments: each row's quad is reloaded for the
                // next k-group right behind the four MFMAs that consumed it (256 cycles before its next use), the single B quad
                // alternates between two sets requested a whole k-group ahead
                static_assert(TN == 1, "rolling reload is written for one column tile per wave");
                auto fetch_b = [&](float4& b) {
                    b = xs[xoff + boff[0]];
                    xoff += step_g;
                    if (++gg == gpt) { gg = 0; xoff += next_tap; if (++kw == p.KW) { kw = 0; xoff += next_row; } }
                };
                auto step = [&](float4& b, float4& bn, int s) {
                    const bool more = s + 1 < ngroups;
                    if (more) fetch_b(bn);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i].x, b.x, acc[i][0], 0, 0, 0);
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i].y, b.y, acc[i][0], 0, 0, 0);
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i].z, b.z, acc[i][0], 0, 0, 0);
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i].w, b.w, acc[i][0], 0, 0, 0);
                        if (more) a0[i] = wt[(s + 1) * 2 * BM + i * 32];
                    }
                };
                fetch_b(b0[0]);
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[i] = wt[i * 32];
                int s = 0;
                for (; s + 2 <= ngroups; s += 2) {
                    step(b0[0], b1[0], s);
                    step(b1[0], b0[0], s + 1);
                }
                if (s < ngroups) step(b0[0], b1[0], s);
            }
            tap0 += p.TT;
            if (tap0 >= p.taps) { tap0 = 0; ++c; }
        }
    }
    trace_mark(trace_wg, 2);
    const bool interior = m_base + BM <= p.Cout_g && h0 + p.TH <= p.Ho && w0 + p.TW <= p.Wo;
    if (!GEN && interior && p.wide_ok) {
        lds_barrier();   // consumers only (the producers have exited): every wave is done with the last stage, LDS is free
        ws_epilogue32_wide<TM, TN>(p, acc, n, g, m_base + wm * (TM * 32), wn * (TN * 32), h0, w0, lane, smem + wave * kEpiScratch);
    } else {
        ws_epilogue32<TM, TN, GEN>(p, acc, n, g, m_base + wm * (TM * 32), wn * (TN * 32), h0, w0, l31, half, interior);
    }
    trace_mark(trace_wg, 3);
    trace_val(trace_wg, 4, (unsigned long long)nstages);
}

// returns 0 launched, < 0 error, 1 the configuration does not fit this form
template <int BM, int BN, int WM, int WN, int KS>
static int launch_conv_ws3(ConvArgs& p, hipStream_t stream) {
    if (p.Cin_g < 8 || !p.w3) return 1;
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
    while (p.BKC > 8 && (p.BKC * p.CHS > 12 * 256 || p.BKC >= 2 * p.Cin_g)) p.BKC >>= 1;
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
    if (p.xs_total > 12 * 256) return 1;
    const int xq = idiv_up(p.xs_total / 4, 256) <= 2 ? 2 : 3;
    const size_t lds = (size_t)(2 * xq * 256 * 4 + 2 * Ws3Geom<BM, KS>::WS_ELEMS) * sizeof(float);
    // 32-bit byte offsets below kBufOob: a channel chunk (plus the 6 channels a quad reaches past its first) and a group of packed
    // weights must span < 2^31 bytes
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
#ifndef AICG_EMULATED
    {
        AICG_SWITCH(stag, "AICG_CONV_STAGGER", 0);
        const long nwg = gx * idiv_up(p.Cout_g, BM) * p.groups;
        const int per_cu = (int)((160 * 1024) / lds);
        if (stag && WM * WN == 4 && per_cu == 2 && nwg >= 8L * 512) {   // >= 8 rounds: the one-off delay costs < 1/16 of the launch
            const long groups_k = (long)p.nchunk * p.taps * (p.BKC / 8);
            const long floor_cycles = groups_k * 4 * (BM / (32 * WM)) * (BN / (32 * WN)) * 64;   // MFMA issue time of one tile
            p.stagger = (int)(lmin(floor_cycles, 1L << 22) * stag / 100);
            p.stagger_first = 512;
        }
    }
#endif
    auto kern = gen ? (xq == 2 ? conv_ws3_kernel<BM, BN, WM, WN, 2, KS, true> : conv_ws3_kernel<BM, BN, WM, WN, 3, KS, true>)
                    : (xq == 2 ? conv_ws3_kernel<BM, BN, WM, WN, 2, KS, false> : conv_ws3_kernel<BM, BN, WM, WN, 3, KS, false>);
    allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return check_launch("conv_ws3_kernel");
}


// ---- 16x16x4 form for 48- / 16-row layers (MDX-Net level 0, RMVPE level 0) -------------------------------------------------------
// v_mfma_f32_16x16x4_f32: lane (r16, q) supplies A[m = r16][k-slot q] and B[k-slot q][n = r16] of a 4-row k-step.  WHICH four
// rows of K form a k-step is free as long as both operands agree, so the k8-interleaved layouts above serve unchanged: of a
// 16-row K group (= two 8-groups, four (group, parity) planes) lane q reads plane q; element u of its float4 is row
// 8 (q >> 1) + 2 u + (q & 1), and k-step u contracts the four rows {2 u, 2 u + 1, 8 + 2 u, 9 + 2 u}.  One ds_read_b128 per fragment
// per FOUR k-steps here too.  Every consumer wave covers all BM rows x 64 positions of a 256-position tile.
template <int BM, int XQ, int KS, bool GEN>
__global__ void __launch_bounds__(512, 4) conv_ws3m16_kernel(ConvArgs p) {
    constexpr int CNT = 256;
    constexpr int TM = BM / 16, TN = 4;
    constexpr int WS_ELEMS = Ws3Geom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = XQ * 256 * 4;
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
        ws3_produce<BM, XQ, KS, false>(p, xs0, ws0, tid - CNT, n, g, h0, w0, m_base, nstages);
        return;
    }
    const int lane = tid & 63, wn = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    int boff[TN];   // float4 index inside a 16-row group's four planes
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * 64 + j * 16 + r16;
        const int jh = nl >> p.TWlog2, jw = nl & (p.TW - 1);
        boff[j] = jh * p.sh * p.TWp + jw * p.sw + q * p.CHS;
    }
    f32x4 acc[TM][TN];
    ws_init_acc16<TM, TN>(p, acc, g, m_base, q);
    const int a_off = q * BM + r16;
    {
        int c = 0, tap0 = 0;
        const int gpt = p.BKC >> 4;   // 16-row groups per tap
        for (int st = 0; st < nstages; ++st) {
            lds_barrier();  // stage st is in LDS
            const float4* xs = reinterpret_cast<const float4*>(xs0 + (c & 1) * XS_ELEMS);
            const float4* wt = reinterpret_cast<const float4*>(ws0 + (st & 1) * WS_ELEMS) + a_off;
            const int nt = imin(p.TT, p.taps - tap0);
            const int ngroups = nt * gpt;
            const int kh0 = tap0 / p.KW;
            int kw = tap0 - kh0 * p.KW, gg = 0;
            int xoff = kh0 * p.dh * p.TWp + kw * p.dw;
            const int step_g = 4 * p.CHS, next_tap = p.dw - gpt * 4 * p.CHS, next_row = p.dh * p.TWp - p.KW * p.dw;
            float4 a0[TM], b0[TN], a1[TM], b1[TN];
            auto fetch = [&](float4 (&a)[TM], float4 (&b)[TN], int s) {
                const float4* xt = xs + xoff;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wt[s * 4 * BM + i * 16];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = xt[boff[j]];
                xoff += step_g;
                if (++gg == gpt) { gg = 0; xoff += next_tap; if (++kw == p.KW) { kw = 0; xoff += next_row; } }
            };
            auto mma = [&](float4 (&a)[TM], float4 (&b)[TN]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = u == 0 ? a[i].x : u == 1 ? a[i].y : u == 2 ? a[i].z : a[i].w;
                            const float bv = u == 0 ? b[j].x : u == 1 ? b[j].y : u == 2 ? b[j].z : b[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
                        }
            };
            fetch(a0, b0, 0);
            int s = 0;
            for (; s + 2 <= ngroups; s += 2) {
                fetch(a1, b1, s + 1);
                mma(a0, b0);
                if (s + 2 < ngroups) fetch(a0, b0, s + 2);  // a 16-row group past the stage would overrun the 8 slack rows
                mma(a1, b1);
            }
            if (s < ngroups) mma(a0, b0);
            tap0 += p.TT;
            if (tap0 >= p.taps) { tap0 = 0; ++c; }
        }
    }
    const bool interior = m_base + BM <= p.Cout_g && h0 + p.TH <= p.Ho && w0 + p.TW <= p.Wo;
    ws_epilogue16<TM, TN, GEN>(p, acc, n, g, m_base, wn * 64, h0, w0, r16, q, interior);
}

template <int BM>
static int launch_conv_ws3m16(ConvArgs& p, hipStream_t stream) {
    constexpr int BN = 256, KS = KSTAGE;
    if (p.Cin_g < 16 || !p.w3) return 1;
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
    // patch budget: 6 quads per producer thread (24 KB per buffer) -- a 256-position 2-D tile with its halo is ~350 positions
    // and a k-step group needs 16 channels of it
    while (p.BKC > 16 && (p.BKC * p.CHS > 24 * 256 || p.BKC >= 2 * p.Cin_g)) p.BKC >>= 1;
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
    if (p.xs_total > 24 * 256) return 1;
    const int xq = idiv_up(p.xs_total / 4, 256) <= 3 ? 3 : 6;
    const size_t lds = (size_t)(2 * xq * 256 * 4 + 2 * Ws3Geom<BM, KS>::WS_ELEMS) * sizeof(float);
    const bool off_ok = (long)(p.BKC + 8) * p.x_sc + (long)p.H * p.x_sh < (1L << 29) && (long)p.taps * p.Cin_pad * p.Mpad < (1L << 29);
    if (lds > 160 * 1024 || !off_ok || (long)p.xs_total * p.CHS >= (1L << 32)) return 1;
    const long gx = (long)p.N * p.tiles_h * p.tiles_w;
    if (gx > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    dim3 grid((unsigned)gx, (unsigned)idiv_up(p.Cout_g, BM), (unsigned)p.groups);
    const bool gen = p.shuffle || p.res_mul;
    p.stagger = p.stagger_first = 0;
    auto kern = gen ? (xq == 3 ? conv_ws3m16_kernel<BM, 3, KS, true> : conv_ws3m16_kernel<BM, 6, KS, true>)
                    : (xq == 3 ? conv_ws3m16_kernel<BM, 3, KS, false> : conv_ws3m16_kernel<BM, 6, KS, false>);
    allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);
    return check_launch("conv_ws3m16_kernel");
}

// ---- 16x16x4 form on 8-byte fragments (48- / 16-row layers, r2) ------------------------------------------------------------------
// The 8-row K groups and the (group, parity) planes of conv_ws3_kernel serve the 16x16x4 MFMA as well: lane (r16, q) takes plane
// parity = q & 1 and of its quad the two elements 2 (q >> 1), 2 (q >> 1) + 1, i.e. rows 8 g + 4 (q >> 1) + 2 u + (q & 1) for
// k-step u = 0, 1 -- four distinct rows per k-step, all eight covered: ONE ds_read_b64 per fragment per TWO k-steps, no larger
// patch than conv_ws3_kernel's (the 16-byte form above needs 16-channel groups and lost to the 4-byte kernel for that reason).
// Halves the LDS -> MFMA hand-overs of conv_ws16_kernel (one ~50-64-cycle bubble per 12 MFMAs of 32 cycles there).
template <int BM, int XQ, int KS, bool GEN>
__global__ void __launch_bounds__(512, 4) conv_ws3m16h_kernel(ConvArgs p) {
    constexpr int CNT = 256;
    constexpr int TM = BM / 16, TN = 4;
    constexpr int WS_ELEMS = Ws3Geom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = XQ * 256 * 4;
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
        ws3_produce<BM, XQ, KS, false>(p, xs0, ws0, tid - CNT, n, g, h0, w0, m_base, nstages);
        return;
    }
    const int lane = tid & 63, wn = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const int par = q & 1, hq = q >> 1;
    int boff[TN];   // float2 index inside a (group, parity) plane pair: 2 x (float4 index) + half
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * 64 + j * 16 + r16;
        const int jh = nl >> p.TWlog2, jw = nl & (p.TW - 1);
        boff[j] = 2 * (jh * p.sh * p.TWp + jw * p.sw + par * p.CHS) + hq;
    }
    f32x4 acc[TM][TN];
    ws_init_acc16<TM, TN>(p, acc, g, m_base, q);
    const int a_off = 2 * (par * BM + r16) + hq;   // float2 index inside a slab pair
    {
        int c = 0, tap0 = 0;
        const int gpt = p.BKC >> 3;   // 8-row groups per tap
        for (int st = 0; st < nstages; ++st) {
            lds_barrier();  // stage st is in LDS
            const float2* xs = reinterpret_cast<const float2*>(xs0 + (c & 1) * XS_ELEMS);
            const float2* wt = reinterpret_cast<const float2*>(ws0 + (st & 1) * WS_ELEMS) + a_off;
            const int nt = imin(p.TT, p.taps - tap0);
            const int ngroups = nt * gpt;
            const int kh0 = tap0 / p.KW;
            int kw = tap0 - kh0 * p.KW, gg = 0;
            int xoff = 2 * (kh0 * p.dh * p.TWp + kw * p.dw);   // float2 units
            const int step_g = 4 * p.CHS, next_tap = 2 * p.dw - gpt * 4 * p.CHS, next_row = 2 * (p.dh * p.TWp - p.KW * p.dw);
            float2 a0[TM], b0[TN], a1[TM], b1[TN];
            auto fetch = [&](float2 (&a)[TM], float2 (&b)[TN], int s) {
                const float2* xt = xs + xoff;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wt[s * 4 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = xt[boff[j]];
                xoff += step_g;
                if (++gg == gpt) { gg = 0; xoff += next_tap; if (++kw == p.KW) { kw = 0; xoff += next_row; } }
            };
            auto mma = [&](float2 (&a)[TM], float2 (&b)[TN]) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u == 0 ? a[i].x : a[i].y, u == 0 ? b[j].x : b[j].y, acc[i][j], 0, 0, 0);
            };
            fetch(a0, b0, 0);
            int s = 0;
            for (; s + 2 <= ngroups; s += 2) {
                fetch(a1, b1, s + 1);
                mma(a0, b0);
                fetch(a0, b0, s + 2);  // unconditional: past the last group this reads (never uses) the LDS slack rows
                mma(a1, b1);
            }
            if (s < ngroups) mma(a0, b0);
            tap0 += p.TT;
            if (tap0 >= p.taps) { tap0 = 0; ++c; }
        }
    }
    const bool interior = m_base + BM <= p.Cout_g && h0 + p.TH <= p.Ho && w0 + p.TW <= p.Wo;
    if (!GEN && interior && p.wide_ok) {
        lds_barrier();   // consumers only: every wave is done with the last stage, LDS is free
        ws_epilogue16_wide<TM, TN>(p, acc, n, g, m_base, wn * 64, h0, w0, lane, smem + wn * kEpi16Scratch);
    } else {
        ws_epilogue16<TM, TN, GEN>(p, acc, n, g, m_base, wn * 64, h0, w0, r16, q, interior);
    }
}

template <int BM>
static int launch_conv_ws3m16h(ConvArgs& p, hipStream_t stream) {
    // KS = 72 K rows per weight stage = all nine taps of a 3 x 3 layer's 8-channel chunk (these layers' patches fit 8 channels): one
    // stage per chunk -- 9 k-groups, ~6 900 MFMA cycles -- instead of two of 5 + 4 with KS = 64: half the stage barriers, and a
    // producer's loads get a whole chunk's MFMAs (> one HBM round trip under load) to land
    // (48-row layers: MDX level 0 118 -> 121 TFLOP/s; the 16-row kernel -- RMVPE level 0, one float4 weight copy per producer
    //  thread at 64 rows, two at 72 -- measured slower with it and keeps 64)
    constexpr int BN = 256, KS = BM == 48 ? 72 : 64;
    if (p.Cin_g < 8 || !p.w3) return 1;
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
    while (p.BKC > 8 && (p.BKC * p.CHS > 12 * 256 || p.BKC >= 2 * p.Cin_g)) p.BKC >>= 1;
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
    if (p.xs_total > 12 * 256) return 1;
    const int xq = idiv_up(p.xs_total / 4, 256) <= 2 ? 2 : 3;
    const size_t lds = (size_t)(2 * xq * 256 * 4 + 2 * Ws3Geom<BM, KS>::WS_ELEMS) * sizeof(float);
    const bool off_ok = (long)(p.BKC + 8) * p.x_sc + (long)p.H * p.x_sh < (1L << 29) && (long)p.taps * p.Cin_pad * p.Mpad < (1L << 29);
    if (lds > 160 * 1024 || !off_ok || (long)p.xs_total * p.CHS >= (1L << 32)) return 1;
    const long gx = (long)p.N * p.tiles_h * p.tiles_w;
    if (gx > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    dim3 grid((unsigned)gx, (unsigned)idiv_up(p.Cout_g, BM), (unsigned)p.groups);
    const bool gen = p.shuffle || p.res_mul;
    p.stagger = p.stagger_first = 0;
    {
        AICG_SWITCH(wide, "AICG_CONV_WIDE", 1);
        p.wide_ok = wide && (size_t)4 * kEpi16Scratch * sizeof(float) <= lds ? conv_wide_ok(p) : 0;
    }
    auto kern = gen ? (xq == 2 ? conv_ws3m16h_kernel<BM, 2, KS, true> : conv_ws3m16h_kernel<BM, 3, KS, true>)
                    : (xq == 2 ? conv_ws3m16h_kernel<BM, 2, KS, false> : conv_ws3m16h_kernel<BM, 3, KS, false>);
    allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);
    return check_launch("conv_ws3m16h_kernel");
}

// instantiation units (conv_ws3_*.hip)
int run_ws3_128x128(ConvArgs& p, hipStream_t st);   // 4 consumers x (128 x 32), 32-row stages
int run_ws3_96x128(ConvArgs& p, hipStream_t st);    // 4 consumers x (96 x 32)
int run_ws3_160x128(ConvArgs& p, hipStream_t st);   // 4 consumers x (160 x 32), one A fragment set reloaded row by row
int run_ws3_64x128(ConvArgs& p, hipStream_t st);    // 4 consumers (2 x 2) x (32 x 64)
int run_ws3_64x64(ConvArgs& p, hipStream_t st);     // 4 consumers (2 x 2) x (32 x 32)
int run_ws3_32x256(ConvArgs& p, hipStream_t st);    // 4 consumers x (32 x 64)
int run_ws3_32x128(ConvArgs& p, hipStream_t st);    // 4 consumers x (32 x 32)
int run_ws3m16_48(ConvArgs& p, hipStream_t st);     // 16x16x4 tiles, 48 rows x 256 positions
int run_ws3m16_16(ConvArgs& p, hipStream_t st);     //                16 rows x 256 positions
int run_ws3m16h_48(ConvArgs& p, hipStream_t st);    // 16x16x4 tiles on 8-byte fragments, 48 rows x 256 positions
int run_ws3m16h_16(ConvArgs& p, hipStream_t st);    //                                     16 rows x 256 positions

}  // namespace aicg
