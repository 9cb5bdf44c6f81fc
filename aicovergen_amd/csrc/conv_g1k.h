// k-tap one-dimensional convolution (unit stride, any dilation, one group, output length = input length) on the machinery of conv_g1.h:
// both operands by LDS DMA, single-role waves, four consecutive positions of a channel per B fragment = four MFMA tiles.  The layers:
// the NSF-HiFiGAN ResBlocks of the vocoder (reference src/infer_pack/modules.py:299-312: k = 3 / 7 / 11, dilations 1 / 3 / 5, 32-256 channels,
// leaky ReLU in front, residual / accumulate behind) -- VERDICT r3 item 4 asked for fewer multiply-adds there (1-D Winograd); this is the
// other route, a busier pipe on the layers the producer / consumer kernel runs below 110 TFLOP/s (256 channels, the k = 3 layers).
//
// K runs over UNITS (channel stage cs of 16 channels, tap t): the weights of a unit are four contiguous slabs of the k8-interleaved image
// ([tap][K / 8][parity][Mpad][4]: tap-major, so a unit looks exactly like a stage of conv_g1) in a ring of three LDS buffers; the input
// WINDOW of a channel stage -- 16 rows of the BN + (k - 1) d positions the tile's outputs reach, starting on a 16-byte boundary -- is
// staged once per channel stage (two buffers) and serves all k taps: tap t reads it shifted by sigma = t d + (window offset) = 4 a + b
// positions.  A lane's four consecutive positions then straddle two quads unless b = 0: it reads quads a + l and a + l + 1 of the row
// (two ds_read_b128) and picks its four values with nine selects on the wave-uniform b, under the MFMAs of the k-step in front.
// Pipeline, epilogue, zero padding (window quads before position 0 or past the end read as zeros through the buffer range check) as in
// conv_g1.h; the window of channel stage cs + 1 is issued behind the barrier in the middle of the FIRST unit of stage cs, in front of the
// weight pieces of unit u + 2, so that the vmcnt(0) of the next barrier covers both in issue order.
#pragma once
#include "conv_g1.h"

namespace aicg {

// positions b .. b + 3 of the eight in (q0, q1), b wave-uniform: two levels of selects (by b & 2, then by b & 1), nine v_cndmask on a
// scalar condition -- under the four MFMAs x TM of the k-step in front.  (Four instantiations of the unit body, one per b, with the
// selection done by register name, made hipcc keep the accumulators in scratch memory: 3 000 spills.)
__device__ __forceinline__ float4 g1k_pick(const float4& q0, const float4& q1, bool b2, bool b1) {
    const float r0 = b2 ? q0.z : q0.x, r1 = b2 ? q0.w : q0.y, r2 = b2 ? q1.x : q0.z, r3 = b2 ? q1.y : q0.w, r4 = b2 ? q1.z : q1.x;
    return make_float4(b1 ? r1 : r0, b1 ? r2 : r1, b1 ? r3 : r2, b1 ? r4 : r3);
}

// p.TWp = quads per window row (RQ), p.div_twp = its reciprocal multiplier; p.taps = k, p.dw = dilation, p.pw = left padding
template <int TM, int WM, int WN, int WPS, bool PRE>
__global__ void __launch_bounds__(256) AICG_WAVES_PER_SIMD(WPS) conv_g1k_kernel(ConvArgs p) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 32 * TM * WM, BN = 128 * WN;
    constexpr int ASTAGE = kG1KS * BM;                   // floats of a unit's weights
    constexpr int NA = BM / 16;
    static_assert(NA % 4 == 0, "every wave issues the same number of weight pieces");
    constexpr int PA = NA / 4;
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* const smem = reinterpret_cast<float*>(smem4);
    const int RQ = p.TWp;                                // quads per window row
    const int BSTAGE = (kG1KS * RQ * 4 + 255) & ~255;    // floats of a channel stage's window, whole DMA pieces (a piece's tail lanes deposit zeros)
    float* const bbuf = smem + 3 * ASTAGE;               // two window buffers behind the weight ring

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid % p.tiles_h;
    const int ct = (bid / p.tiles_h) % p.tiles_w;
    const int img = bid / (p.tiles_h * p.tiles_w);
    const int m_base = mt * BM;
    const int T = p.Wo;
    const int n0 = ct * BN;
    const int nst = (p.Cin_g + kG1KS - 1) / kG1KS;
    const int ntap = p.taps;
    const int nunits = nst * ntap;
    // the window: input positions s0 .. s0 + 4 RQ - 1, s0 = the 16-byte boundary at or below n0 - pw; tap t reads it shifted by t d + delta0
    const int first_in = n0 - p.pw;
    const int s0 = first_in & ~3;                        // (two's complement: floor to a multiple of 4 for negative values too)
    const int delta0 = first_in - s0;

    unsigned aoff[PA];
#pragma unroll
    for (int e = 0; e < PA; ++e) {
        const int q = (wave + 4 * e) * 64 + lane;
        const int slab = q / BM, m = q - slab * BM;
        aoff[e] = m_base + m < p.Mpad ? 16u * (unsigned)(slab * p.Mpad + m_base + m) : kBufOob;
    }
    const float* const ximg = p.x + (long)img * p.x_sn;
    const long wtap = (long)(p.Cin_pad >> 3) * 2 * p.Mpad * 4;       // floats of one tap's k8-interleaved image
    BufRsrc wb;
    auto unit_rsrc = [&](int cs, int t) __attribute__((always_inline)) {
        const long wbase = (long)t * wtap + (long)cs * (kG1KS / 8) * 2 * p.Mpad * 4;
        wb = make_buf(p.w3 + wbase, (unsigned)lmin((wtap - (long)cs * (kG1KS / 8) * 2 * p.Mpad * 4) * 4, 0x7fffffffL));
    };
    auto issue_a_piece = [&](int e, float* abuf) __attribute__((always_inline)) {
        w2d_dma16(wb, aoff[e], 0u, abuf + (wave + 4 * e) * 256, lane);
    };
    auto issue_a = [&](int cs, int t, float* abuf) __attribute__((always_inline)) {
        unit_rsrc(cs, t);
#pragma unroll
        for (int e = 0; e < PA; ++e) issue_a_piece(e, abuf);
    };
    // the window of channel stage cs: 16 rows x RQ quads, lane-linear; quads before position 0 / past the end / of absent channels read 0
    auto issue_b = [&](int cs, float* dst) __attribute__((always_inline)) {
        const long left = (long)(p.Cin_g - cs * kG1KS) * p.x_sc * 4;
        const BufRsrc xb = make_buf(ximg + (long)cs * kG1KS * p.x_sc, (unsigned)lmin(left, 0x7fffffffL));
        const int nq = kG1KS * RQ;
        for (int piece = wave; piece * 64 < nq; piece += 4) {
            const int q = piece * 64 + lane;
            const int row = (int)__umulhi((unsigned)q, p.div_twp);
            const int col = q - row * RQ;
            const int pos = s0 + 4 * col;
            const bool ok = q < nq && pos >= 0 && pos < T;
            w2d_dma16(xb, ok ? 4u * (unsigned)(row * (int)p.x_sc + pos) : kBufOob, 0u, dst + piece * 256, lane);
        }
    };

    f32x16 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_lane = half * BM + wm * (TM * 32) + l31;
    const int b_lane = half * RQ + wn * 32 + l31;        // float4 index of this lane's quad at shift 0 inside a row pair
    // Fragments.  Weights: TM quads per k-group, two sets (a0: first group of a unit, a1: second).  Window: per k-step the two quads the
    // shifted positions straddle, ONE set of 4 x 2 quads that rolls -- right behind the selects that consumed k-step s of a group, its two
    // registers are reloaded with k-step s of the NEXT group (three k-steps = 24 TM MFMAs ahead of their use); a second set would put the
    // kernel past 256 registers (353 spills, the accumulators among them).
    auto read_a = [&](float4 (&a)[TM], const float* abuf, int g) __attribute__((always_inline)) {
        const float4* wt = reinterpret_cast<const float4*>(__builtin_assume_aligned(abuf, 16)) + a_lane + g * 2 * BM;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = wt[i * 32];
    };
    auto window_ptr = [&](const float* wbuf, int g, int qa) __attribute__((always_inline)) {
        return reinterpret_cast<const float4*>(__builtin_assume_aligned(wbuf, 16)) + b_lane + 8 * g * RQ + qa;
    };
    float4 bq0[4], bq1[4];
    const float pre_slope = p.pre_slope;
    // one k-group: per k-step select (+ leaky ReLU), reload of the step's window registers from `nxt` (the next group's quads; nullptr:
    // none), 4 TM MFMAs; `dma`: a weight piece of the unit two ahead between two blocks of four MFMAs (conv_g1.h)
    auto mma_group = [&](int sh, const float4 (&a)[TM], const float4* nxt, auto reload_tag, auto dma_tag, bool more, float* fill) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_tag)::value;
        constexpr bool RELOAD = decltype(reload_tag)::value;
        constexpr int NBLK = 4 * TM;
        const bool sh2 = (sh & 2) != 0, sh1 = (sh & 1) != 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 bv = g1k_pick(bq0[s], bq1[s], sh2, sh1);
            if constexpr (PRE) {   // lrelu(v) = max(v, slope v) for 0 <= slope <= 1
                bv.x = fmaxf(bv.x, bv.x * pre_slope); bv.y = fmaxf(bv.y, bv.y * pre_slope);
                bv.z = fmaxf(bv.z, bv.z * pre_slope); bv.w = fmaxf(bv.w, bv.w * pre_slope);
            }
            w2d_fence();
            if constexpr (RELOAD) { bq0[s] = nxt[2 * s * RQ]; bq1[s] = nxt[2 * s * RQ + 1]; }
            w2d_fence();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float av = s == 0 ? a[i].x : s == 1 ? a[i].y : s == 2 ? a[i].z : a[i].w;
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, acc[i][1], 0, 0, 0);
                acc[i][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, acc[i][2], 0, 0, 0);
                acc[i][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, acc[i][3], 0, 0, 0);
                if constexpr (DMA) {
                    const int blk = s * TM + i;
                    w2d_fence();
                    if (more) {
#pragma unroll
                        for (int e = 0; e < PA; ++e)
                            if (e * NBLK / PA == blk) issue_a_piece(e, fill);
                    }
                    w2d_fence();
                }
            }
        }
    };

    // ---- the unit pipeline (conv_g1.h's stage pipeline with unit = (channel stage, tap))
    float* a_cur = smem;
    float* a_nxt = smem + ASTAGE;
    float* a_fill = smem + 2 * ASTAGE;
    int cs = 0, t = 0;                                   // the running unit
    issue_b(0, bbuf);
    issue_a(0, 0, a_cur);
    if (nunits > 1) issue_a(0, 1, a_nxt);                // (k >= 2: unit 1 is tap 1 of channel stage 0)
    g1_wait_pieces<0>();
    lds_barrier();
    float4 a0[TM], a1[TM];
    int sig = delta0;                                    // shift of the running unit's tap: t d + delta0
    read_a(a0, a_cur, 0);
    {
        const float4* w0 = window_ptr(bbuf, 0, sig >> 2);
#pragma unroll
        for (int s = 0; s < 4; ++s) { bq0[s] = w0[2 * s * RQ]; bq1[s] = w0[2 * s * RQ + 1]; }
    }
    // (the last unit is peeled, as conv_g1.h peels its last stage: no conditional section inside the loop body)
    for (int u = 0; u + 1 < nunits; ++u) {
        const float* wcur = bbuf + (cs & 1) * BSTAGE;
        w2d_fence();
        read_a(a1, a_cur, 1);
        w2d_fence();
        mma_group(sig & 3, a0, window_ptr(wcur, 1, sig >> 2), std::true_type{}, std::false_type{}, false, nullptr);
        w2d_fence();
        g1_wait_pieces<0>();   // unit u + 1's weights (and, issued in front of them, the next channel stage's window)
        lds_barrier();
        // the next unit and the one behind it
        int cs1 = cs, t1 = t + 1;
        if (t1 == ntap) { t1 = 0; ++cs1; }
        int cs2 = cs1, t2 = t1 + 1;
        if (t2 == ntap) { t2 = 0; ++cs2; }
        const bool more = cs2 < nst;
        if (t == 0 && cs + 1 < nst) issue_b(cs + 1, bbuf + ((cs + 1) & 1) * BSTAGE);   // every wave is past channel stage cs - 1
        if (more) unit_rsrc(cs2, t2);
        const int sig1 = t1 * p.dw + delta0;
        read_a(a0, a_nxt, 0);
        w2d_fence();
        mma_group(sig & 3, a1, window_ptr(bbuf + (cs1 & 1) * BSTAGE, 0, sig1 >> 2), std::true_type{}, std::true_type{}, more, a_fill);
        float* tmp = a_cur; a_cur = a_nxt; a_nxt = a_fill; a_fill = tmp;
        cs = cs1; t = t1; sig = sig1;
    }
    {
        w2d_fence();
        read_a(a1, a_cur, 1);
        w2d_fence();
        mma_group(sig & 3, a0, window_ptr(bbuf + (cs & 1) * BSTAGE, 1, sig >> 2), std::true_type{}, std::false_type{}, false, nullptr);
        w2d_fence();
        mma_group(sig & 3, a1, nullptr, std::false_type{}, std::false_type{}, false, nullptr);
    }
    g1_epilogue<TM, false>(p, acc, img, m_base + wm * (TM * 32), n0 + wn * 128 + 4 * l31, T);
}

// quads per window row for a tile of bn outputs: the positions n0 - pw .. n0 + bn - 1 + (k - 1) d - pw from the 16-byte boundary below
// the first (up to 3 positions lower), plus the quad a shifted read reaches past its first
__host__ __device__ constexpr int g1k_row_quads(int bn, int k, int d) { return (bn + (k - 1) * d + 3 + 3) / 4 + 1; }

// host side: a 1-D layer of the form this kernel takes
inline bool conv_g1k_applicable(const ConvArgs& p, int pad_w_end) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    auto m4 = [](long v) { return (v & 3) == 0; };
    if (p.KH != 1 || p.H != 1 || p.Ho != 1 || p.KW < 2 || p.KW > 16 || p.groups != 1 || p.sw != 1 || p.sh != 1 || p.ph || !p.w3) return false;
    if (p.pw + pad_w_end != (p.KW - 1) * p.dw || p.Wo != p.W || (p.W & 3) || p.Cin_g < 16) return false;
    if (p.pre_act != AICG_ACT_NONE && !(p.pre_act == AICG_ACT_LRELU && p.pre_slope >= 0.f && p.pre_slope <= 1.f)) return false;
    if (p.shuffle || p.res_mul || p.W >= (1 << 24) || p.x_sc >= (1L << 24) || p.x_sc < p.W) return false;
    if ((p.KW - 1) * p.dw > 256) return false;
    if (!al(p.x) || !m4(p.x_sn) || !m4(p.x_sc) || !al(p.y) || !m4(p.y_sn) || !m4(p.y_sc)) return false;
    if (p.res && (!al(p.res) || !m4(p.r_sn) || !m4(p.r_sc))) return false;
    return true;
}

template <int TM, int WM, int WN, int WPS>
static int launch_conv_g1k(ConvArgs& p, hipStream_t stream) {
    constexpr int BM = 32 * TM * WM, BN = 128 * WN;
    p.tiles_h = idiv_up(p.Cout_g, BM);
    p.tiles_w = idiv_up(p.Wo, BN);
    const long nwg = (long)p.N * p.tiles_h * p.tiles_w;
    if (nwg > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    p.TWp = g1k_row_quads(BN, p.KW, p.dw);
    p.div_twp = div_mul(p.TWp);
    const size_t lds = (size_t)(3 * kG1KS * BM + 2 * ((kG1KS * p.TWp * 4 + 255) & ~255)) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    auto kern = p.pre_act != AICG_ACT_NONE ? conv_g1k_kernel<TM, WM, WN, WPS, true> : conv_g1k_kernel<TM, WM, WN, WPS, false>;
    allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, stream, p);
    return check_launch("conv_g1k_kernel");
}

// instantiation unit conv_g1k_1.hip
int run_g1k_128x256(ConvArgs& p, hipStream_t st);
int run_g1k_64x256(ConvArgs& p, hipStream_t st);

}  // namespace aicg
