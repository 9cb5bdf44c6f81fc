// 1 x 1 convolution (unit stride, no padding, one group, contiguous maps) as the plain GEMM it is,
//     Y[M, P] = W[M, K] X[K, P]            M = Cout, K = Cin, P = H * W positions of one image,
// with BOTH operands staged by LDS DMA and single-role waves -- what tdf_pair.hip (123-131 TFLOP/s) and conv_w2d.h showed for their shapes,
// on a plain K loop.  VERDICT r3 item 1: the per-token GEMMs of HuBERT (768 <-> 3072 at 13 198 tokens: 66-86 TFLOP/s on the
// producer / consumer kernel of conv_ws3.h), MDX-Net's up-sampling GEMM + pixel shuffle (58-70), the 1 x 1 layers of enc_p / flow / vocoder
// (reference ops: src/infer_pack/attentions.py:190-193,387-388, the fairseq encoder behind src/rvc.py:98-109, src/mdx.py:74-77).
//
// Operand images.  X is channel-major, a channel's positions are contiguous in HBM -- exactly what an LDS-DMA piece wants (64 lanes x
// 16 bytes = 256 consecutive positions of one channel land as 1 KiB of LDS): the B stage is [k][BN] floats, row k = input channel.
// A lane's B fragment is then ONE ds_read_b128 of FOUR CONSECUTIVE POSITIONS of channel k = 2 s + half, and the four values feed four
// different 32 x 32 MFMA tiles: tile j of a wave owns the positions 4 l + j (l = 0..31) of the wave's 128 -- a permutation of the
// columns that costs nothing (the conv_ws3 kernels get the same 16 bytes per four MFMAs from four k-steps of one position and pay for
// it with a producer's register transpose).  It also pays in the epilogue: a lane ends up with four consecutive positions of every
// row it owns -- float4 stores (and float4 residual loads) straight from the accumulators, 512-byte runs per 32 lanes, no LDS detour.
// W comes from the k8-interleaved image the layer already carries (ops.pack_conv_weight: [K / 8][parity][Mpad][4], element j = row
// 8 g + 2 j + parity): a K stage of a BM-row tile is four contiguous slabs, and one ds_read_b128 is a row's A fragments of four k-steps.
//
// Tile: a workgroup of four waves (WM x WN) owns BM = 32 TM WM rows x BN = 128 WN positions; a wave TM x 4 accumulator tiles
// (TM = 2: 128 registers, two workgroups per CU -- one's barrier / epilogue under the other's MFMAs).  K runs in stages of 16 rows
// (two k-groups of 8 = four MFMA k-steps each), THREE LDS buffers; a wave waits for its own DMA pieces, the LDS-only barrier
// publishes them (the pipeline is described at the loop).
// Zero padding -- channels past Cin, positions past the end of the map, rows past Mpad -- is the buffer range check's (offset
// kBufOob, or past num_records).  Input activations other than a leaky ReLU stay on conv_ws3.
#pragma once
#include "conv_w2d.h"

namespace aicg {

static constexpr int kG1KS = 16;     // K rows per stage (two k-groups of 8)
static constexpr int kG1Bufs = 3;

__host__ __device__ constexpr int g1_stage_floats(int bm, int bn) { return kG1KS * (bm + bn); }

template <int N>
__device__ __forceinline__ void g1_wait_pieces() {   // at most N of this wave's DMA pieces still in flight
#ifndef AICG_EMULATED
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// Epilogue shared by conv_g1_kernel and conv_g1k_kernel.  Register r of tile (i, j): row m_wave0 + 32 i + 8 (r >> 2) + 4 half + (r & 3),
// position pos + j (pos = the first of this lane's four consecutive positions; HW % 4 == 0: a quad is inside the map or outside).
// y = [y +] out_scale * (act(acc + bias [+ res]) [+ res | * res]); SHUF: the k = s = 2 ConvTranspose2d scatter.
// RAGGED (conv_g1s.h): HW is any length -- the quad that straddles the end of the row stores its leading elements one by one.
template <int TM, bool SHUF, bool RAGGED = false>
__device__ __forceinline__ void g1_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][4], int img, int m_wave0, int pos, int HW) {
    static_assert(!(SHUF && RAGGED), "the pixel-shuffle scatter works on whole quads");
    const int lane = (int)(threadIdx.x & 63), half = lane >> 5;
    const bool pos_ok = pos < HW;                                     // (HW % 4 == 0: a quad is inside the map or outside; no early exit --
                                                                      //  every lane's bias register is a shuffle source)
    const int nvalid = RAGGED ? (HW - pos < 4 ? HW - pos : 4) : 4;     // elements of this lane's quad inside the row
    // bias: the wave's rows are contiguous -- one coalesced load per 64 rows, every (tile, register) slot picks its value with a shuffle
    constexpr int NBR = (TM * 32 + 63) / 64;
    float breg[NBR];
#pragma unroll
    for (int t = 0; t < NBR; ++t) {
        const int m = m_wave0 + 64 * t + lane;
        breg[t] = (p.bias && m < p.Cout_g) ? p.bias[m] : 0.f;
    }
    auto body = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
        if constexpr (SHUF) {
            // rows 4 co .. 4 co + 3 are the taps (dy, dx) of output channel co; positions (ho, wo .. wo + 3) -> rows 2 ho + dy, columns
            // 2 wo .. 2 wo + 7: two float4 per (channel, dy)
            const int ho = pos / p.Wo, wo = pos - ho * p.Wo;
            const long y_pos = (long)img * p.y_sn + (long)(2 * ho) * p.y_sh + 2 * wo;
            const long r_pos = (long)img * p.r_sn + (long)(2 * ho) * p.r_sh + 2 * wo;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m0 = m_wave0 + i * 32 + 8 * q + 4 * half;
                    float bb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) bb[e] = __shfl(breg[i >> 1], (i & 1) * 32 + 8 * q + 4 * half + e, 64);
                    if (m0 >= p.Cout_g || !pos_ok) continue;
                    const long co = m0 >> 2;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int hj = 0; hj < 2; ++hj) {
                            float v[4] = {acc[i][2 * hj][4 * q + 2 * dy] + bb[2 * dy], acc[i][2 * hj][4 * q + 2 * dy + 1] + bb[2 * dy + 1],
                                          acc[i][2 * hj + 1][4 * q + 2 * dy] + bb[2 * dy], acc[i][2 * hj + 1][4 * q + 2 * dy + 1] + bb[2 * dy + 1]};
                            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (p.res) rv = *reinterpret_cast<const float4*>(p.res + r_pos + co * p.r_sc + (long)dy * p.r_sh + 4 * hj);
                            const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                float x = act_static<ACT>(v[t], p.act, p.act_slope);
                                if (p.res) x = p.res_mul ? x * rr[t] : x + rr[t];
                                v[t] = x * p.out_scale;
                            }
                            *reinterpret_cast<float4*>(p.y + y_pos + co * p.y_sc + (long)dy * p.y_sh + 4 * hj) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                }
        } else {
            const long y_col = (long)img * p.y_sn + pos, r_col = (long)img * p.r_sn + pos;
            // ONE statement of the per-element epilogue (additive residual before / after the activation, output scale, accumulate), used
            // by the float4 path and by the ragged last quad alike.  (res_mul is the SHUF branch's; conv_g1s_applicable() rejects it,
            // so RAGGED never meets it.)
            auto elem = [&](float x, float rr, float yy) __attribute__((always_inline)) {
                if (p.res_first) x += rr;
                x = act_static<ACT>(x, p.act, p.act_slope);
                if (!p.res_first) x += rr;
                return x * p.out_scale + yy;
            };
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_wave0 + i * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                    const float bv = __shfl(breg[i >> 1], (i & 1) * 32 + 8 * (r >> 2) + 4 * half + (r & 3), 64);
                    if (m >= p.Cout_g || !pos_ok) continue;
                    float v[4] = {acc[i][0][r] + bv, acc[i][1][r] + bv, acc[i][2][r] + bv, acc[i][3][r] + bv};
                    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), yv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (RAGGED) {
                        if (nvalid < 4) {   // the row's last, partial quad: element by element
#pragma unroll
                            for (int t = 0; t < 3; ++t) {
                                if (t >= nvalid) break;
                                const float rr1 = p.res ? p.res[r_col + (long)m * p.r_sc + t] : 0.f;
                                float* dst1 = p.y + y_col + (long)m * p.y_sc + t;
                                *dst1 = elem(v[t], rr1, p.accumulate ? *dst1 : 0.f);
                            }
                            continue;
                        }
                    }
                    if (p.res) rv = *reinterpret_cast<const float4*>(p.res + r_col + (long)m * p.r_sc);
                    if (p.accumulate) yv = *reinterpret_cast<const float4*>(p.y + y_col + (long)m * p.y_sc);
                    const float rr[4] = {rv.x, rv.y, rv.z, rv.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = elem(v[t], rr[t], yy[t]);
                    *reinterpret_cast<float4*>(p.y + y_col + (long)m * p.y_sc) = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
    };
    if (p.act == AICG_ACT_NONE) body(std::integral_constant<int, 0>{});
    else if (p.act == AICG_ACT_RELU) body(std::integral_constant<int, 1>{});
    else if (p.act == AICG_ACT_LRELU) body(std::integral_constant<int, 2>{});
    else if (!SHUF && p.act == AICG_ACT_GELU) body(std::integral_constant<int, 4>{});
    else body(std::integral_constant<int, 3>{});
}

// SHUF: the kernel = stride = 2 ConvTranspose2d scatter (ConvArgs::shuffle == 2) with its additive / multiplicative skip operand.
// WPS: waves per SIMD the register allocation is held to.
// SPREAD: the DMA pieces of a stage go out one at a time between the MFMAs (see mma_group) instead of as one burst behind the barrier.
// PRE: a leaky-ReLU input activation (0 <= slope <= 1: the vocoder's up-sampling GEMMs, models.py:503), applied to the B fragments in
// registers right in front of the k-step that consumes them -- 8 VALU operations under the previous k-step's 4 TM MFMAs.
// F16 (aicg_conv_desc.split == 2, the reference's is_half mode): the same staging and the same fp32 images; a k-group's fragments -- an A
// quad = a row's four k-steps, the B quads of four k-steps = 4 x 4 (k-step, position) values -- are rounded to fp16 in registers and
// contracted by ONE v_mfma_f32_32x32x8_f16 per (row tile, position tile) where the fp32 path issues four 32 x 32 x 2 MFMAs: lane half h
// supplies k = 4 h + e <-> channel 2 e + h of the group on both operands.  fp32 accumulation and epilogue.
template <int TM, int WM, int WN, bool SHUF, int WPS, bool SPREAD, bool PRE, bool F16 = false>
__global__ void __launch_bounds__(256) AICG_WAVES_PER_SIMD(WPS) conv_g1_kernel(ConvArgs p) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 32 * TM * WM, BN = 128 * WN;
    constexpr int STAGE = g1_stage_floats(BM, BN);
    constexpr int NA = BM / 16, NB = BN / 16;            // 1 KiB DMA pieces of a stage: weights, positions
    static_assert(NA % 4 == 0 && NB % 4 == 0, "every wave issues the same number of pieces (the counted wait)");
    constexpr int PA = NA / 4, PB = NB / 4, PW = PA + PB;
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* const smem = reinterpret_cast<float*>(smem4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    // tile walk: M tile fastest -- the workgroups that share a slab of X run side by side on one XCD's L2
    const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid % p.tiles_h;
    const int ct = (bid / p.tiles_h) % p.tiles_w;
    const int img = bid / (p.tiles_h * p.tiles_w);
    const int m_base = mt * BM;
    const int HW = p.Ho * p.Wo;
    const int n0 = ct * BN;
    const int nst = (p.Cin_g + kG1KS - 1) / kG1KS;

    // ---- this wave's DMA pieces: byte offsets relative to the stage's base (weights: slab (g, parity) of BM quads; positions: row k)
    unsigned aoff[PA], boff[PB];
#pragma unroll
    for (int e = 0; e < PA; ++e) {
        const int q = (wave + 4 * e) * 64 + lane;         // quad of the stage's A image: [slab][m]
        const int slab = q / BM, m = q - slab * BM;
        aoff[e] = m_base + m < p.Mpad ? 16u * (unsigned)(slab * p.Mpad + m_base + m) : kBufOob;
    }
#pragma unroll
    for (int e = 0; e < PB; ++e) {
        const int q = (wave + 4 * e) * 64 + lane;         // quad of the stage's B image: [k][BN / 4]
        const int k = q / (BN / 4), c4 = q - k * (BN / 4);
        boff[e] = n0 + 4 * c4 < HW ? 4u * (unsigned)(k * (int)p.x_sc + n0 + 4 * c4) : kBufOob;
    }
    const float* const ximg = p.x + (long)img * p.x_sn;
    const long wtotal = (long)(p.Cin_pad >> 3) * 2 * p.Mpad * 4;      // floats of the k8-interleaved image
    // stage st's operand windows (wave-uniform resources) and this wave's pieces of it, one at a time
    BufRsrc wb, xb;
    auto stage_rsrc = [&](int st) __attribute__((always_inline)) {
        const long wbase = (long)st * (kG1KS / 8) * 2 * p.Mpad * 4;
        wb = make_buf(p.w3 + wbase, (unsigned)lmin((wtotal - wbase) * 4, 0x7fffffffL));
        const long left = (long)(p.Cin_g - st * kG1KS) * p.x_sc * 4;   // bytes up to the end of the image's channels: absent channels read 0
        xb = make_buf(ximg + (long)st * kG1KS * p.x_sc, (unsigned)lmin(left, 0x7fffffffL));
    };
    auto issue_piece = [&](int e, float* buf) __attribute__((always_inline)) {   // e < PA: weights, else positions
        if (e < PA) w2d_dma16(wb, aoff[e], 0u, buf + (wave + 4 * e) * 256, lane);
        else w2d_dma16(xb, boff[e - PA], 0u, buf + kG1KS * BM + (wave + 4 * (e - PA)) * 256, lane);
    };
    auto issue = [&](int st, float* buf) __attribute__((always_inline)) {
        stage_rsrc(st);
#pragma unroll
        for (int e = 0; e < PW; ++e) issue_piece(e, buf);
    };

    f32x16 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_lane = half * BM + wm * (TM * 32) + l31;            // float4 index inside a k-group's slab pair
    const int b_lane = half * (BN / 4) + wn * 32 + l31;            // float4 index inside a row pair of the B image
    // fragments of k-group g (8 rows = 4 k-steps) of the stage at `stage`: TM + 4 ds_read_b128 for 16 TM MFMAs
    auto read_group = [&](float4 (&a)[TM], float4 (&b)[4], const float* stage, int g) __attribute__((always_inline)) {
        const float4* wt = reinterpret_cast<const float4*>(__builtin_assume_aligned(stage, 16)) + a_lane + g * 2 * BM;
        const float4* xt = reinterpret_cast<const float4*>(__builtin_assume_aligned(stage + kG1KS * BM, 16)) + b_lane + 8 * g * (BN / 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = wt[i * 32];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = xt[2 * s * (BN / 4)];
    };
    // `dma`: this group also carries the wave's DMA pieces of the stage two ahead, ONE PIECE BETWEEN TWO BLOCKS OF FOUR MFMAs -- a piece
    // costs the wave ~60-180 cycles of issue (M0 juggling + the buffer instruction); in one burst behind the barrier that was 10-15 %
    // of a stage with the matrix pipe idle, spread out it disappears under the 256 cycles of the block in front of it
    const float pre_slope = p.pre_slope;
    auto mma_group = [&](const float4 (&a)[TM], const float4 (&b)[4], auto dma_tag, bool more, float* fill) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_tag)::value;
        constexpr int NBLK = 4 * TM;
        if constexpr (F16) {
            // (scalars, not a copied float4 array: hipcc put that one in scratch memory)
            auto pre = [&](float v) __attribute__((always_inline)) { return PRE ? fmaxf(v, v * pre_slope) : v; };
            const H4 bh0 = pack_f16x4(pre(b[0].x), pre(b[1].x), pre(b[2].x), pre(b[3].x));
            const H4 bh1 = pack_f16x4(pre(b[0].y), pre(b[1].y), pre(b[2].y), pre(b[3].y));
            const H4 bh2 = pack_f16x4(pre(b[0].z), pre(b[1].z), pre(b[2].z), pre(b[3].z));
            const H4 bh3 = pack_f16x4(pre(b[0].w), pre(b[1].w), pre(b[2].w), pre(b[3].w));
            auto piece_behind = [&](int blk) __attribute__((always_inline)) {
                if constexpr (DMA) {
                    w2d_fence();
                    if (more) {
#pragma unroll
                        for (int e = 0; e < PW; ++e)
                            if (e * NBLK / PW == blk) issue_piece(e, fill);
                    }
                    w2d_fence();
                }
            };
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const H4 ah = pack_f16x4(a[i].x, a[i].y, a[i].z, a[i].w);
                acc[i][0] = mfma_f16_32x32x8(ah, bh0, acc[i][0]); piece_behind(i * 4 + 0);
                acc[i][1] = mfma_f16_32x32x8(ah, bh1, acc[i][1]); piece_behind(i * 4 + 1);
                acc[i][2] = mfma_f16_32x32x8(ah, bh2, acc[i][2]); piece_behind(i * 4 + 2);
                acc[i][3] = mfma_f16_32x32x8(ah, bh3, acc[i][3]); piece_behind(i * 4 + 3);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float av = s == 0 ? a[i].x : s == 1 ? a[i].y : s == 2 ? a[i].z : a[i].w;
                float4 bv = b[s];
                if constexpr (PRE) {   // lrelu(v) = max(v, slope v) for 0 <= slope <= 1 (same bits for every finite v)
                    bv.x = fmaxf(bv.x, bv.x * pre_slope); bv.y = fmaxf(bv.y, bv.y * pre_slope);
                    bv.z = fmaxf(bv.z, bv.z * pre_slope); bv.w = fmaxf(bv.w, bv.w * pre_slope);
                }
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, acc[i][1], 0, 0, 0);
                acc[i][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, acc[i][2], 0, 0, 0);
                acc[i][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, acc[i][3], 0, 0, 0);
                if constexpr (DMA) {
                    const int blk = s * TM + i;
                    w2d_fence();
                    if (more) {
#pragma unroll
                        for (int e = 0; e < PW; ++e)
                            if (e * NBLK / PW == blk) issue_piece(e, fill);
                    }
                    w2d_fence();
                }
            }
    };

    // One software pipeline over the k-groups of all stages: while a group's MFMAs run, the next group's fragments are on their way
    // from LDS -- also across the stage boundary, because the barrier that publishes stage s + 1 sits in the MIDDLE of stage s (between
    // its two groups): behind it stage s + 1 is complete, every wave is done with stage s - 1, whose buffer takes the DMA of stage
    // s + 2 (a stage and a half = 96 TM MFMAs of lead), and the second group of stage s prefetches the first of stage s + 1.
    float* b_cur = smem;
    float* b_nxt = smem + STAGE;
    float* b_fill = smem + 2 * STAGE;
    issue(0, b_cur);
    if (nst > 1) { issue(1, b_nxt); g1_wait_pieces<PW>(); } else g1_wait_pieces<0>();
    lds_barrier();
    float4 a0[TM], b0[4], a1[TM], b1[4];
    read_group(a0, b0, b_cur, 0);
    // (the last stage is peeled: a conditional mid-stage section would merge two LDS-counter states at its end, and hipcc then guards the
    //  second group's MFMAs with waits for the fragment reads issued a moment ago)
    for (int st = 0; st + 1 < nst; ++st) {
        w2d_fence();
        read_group(a1, b1, b_cur, 1);
        w2d_fence();
        mma_group(a0, b0, std::false_type{}, false, nullptr);
        w2d_fence();
        g1_wait_pieces<0>();   // this wave's pieces of stage st + 1 (the only ones in flight)
        lds_barrier();
        const bool more = st + 2 < nst;
        if constexpr (SPREAD) { if (more) stage_rsrc(st + 2); } else { if (more) issue(st + 2, b_fill); }
        read_group(a0, b0, b_nxt, 0);
        w2d_fence();
        mma_group(a1, b1, std::integral_constant<bool, SPREAD>{}, more, b_fill);
        float* t = b_cur; b_cur = b_nxt; b_nxt = b_fill; b_fill = t;
    }
    w2d_fence();
    read_group(a1, b1, b_cur, 1);
    w2d_fence();
    mma_group(a0, b0, std::false_type{}, false, nullptr);
    w2d_fence();
    mma_group(a1, b1, std::false_type{}, false, nullptr);

    g1_epilogue<TM, SHUF>(p, acc, img, m_base + wm * (TM * 32), n0 + wn * 128 + 4 * l31, HW);
}

// host side: does the layer have the form this kernel takes?  (1 x 1, unit stride, no padding, one group, no input activation
// but a leaky ReLU, contiguous 16-byte-aligned maps whose size is a multiple of 4)
inline bool conv_g1_applicable(const ConvArgs& p, int pad_h_end, int pad_w_end) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    auto m4 = [](long v) { return (v & 3) == 0; };
    if (p.taps != 1 || p.groups != 1 || p.sh != 1 || p.sw != 1 || p.ph || p.pw || pad_h_end || pad_w_end || !p.w3) return false;
    if (p.pre_act != AICG_ACT_NONE && !(p.pre_act == AICG_ACT_LRELU && p.pre_slope >= 0.f && p.pre_slope <= 1.f && !p.shuffle)) return false;
    if (p.Ho != p.H || p.Wo != p.W || p.Cin_g < 8) return false;
    const long HW = (long)p.H * p.W;
    if ((HW & 3) || HW >= (1L << 24) || p.x_sc >= (1L << 24) || (p.Cin_g > 1 && p.x_sc < HW)) return false;   // 32-bit byte offsets of a 16-row stage
    if (p.H > 1 && p.x_sh != p.W) return false;                                // a channel's positions are one contiguous run
    if (!al(p.x) || !m4(p.x_sn) || !m4(p.x_sc) || !al(p.y) || !m4(p.y_sn) || !m4(p.y_sc)) return false;
    if (p.res && (!al(p.res) || !m4(p.r_sn) || !m4(p.r_sc))) return false;
    if (p.shuffle) {
        if (p.shuffle != 2 || p.accumulate || p.res_first || (p.W & 3) || (p.Cout_g & 3) || !m4(p.y_sh) || (p.res && !m4(p.r_sh))) return false;
    } else {
        if (p.res_mul) return false;
        if (p.H > 1 && (p.y_sh != p.Wo || (p.res && p.r_sh != p.Wo))) return false;
    }
    return true;
}

template <int TM, int WM, int WN, int WPS, bool SPREAD = true, bool F16 = false>
static int launch_conv_g1(ConvArgs& p, hipStream_t stream) {
    constexpr int BM = 32 * TM * WM, BN = 128 * WN;
    const long HW = (long)p.H * p.W;
    p.tiles_h = idiv_up(p.Cout_g, BM);
    p.tiles_w = (int)ldiv_up(HW, BN);
    const long nwg = (long)p.N * p.tiles_h * p.tiles_w;
    if (nwg > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    const size_t lds = (size_t)kG1Bufs * g1_stage_floats(BM, BN) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    if (F16 && p.shuffle) return 1;                  // (the pixel-shuffle layers are MDX-Net's: no is_half mode there)
    auto kern = p.shuffle ? conv_g1_kernel<TM, WM, WN, true, WPS, SPREAD, false>
                          : p.pre_act != AICG_ACT_NONE ? conv_g1_kernel<TM, WM, WN, false, WPS, SPREAD, true, F16> : conv_g1_kernel<TM, WM, WN, false, WPS, SPREAD, false, F16>;
    allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, stream, p);
    return check_launch("conv_g1_kernel");
}

// instantiation units (conv_g1_*.hip)
int run_g1_128x256(ConvArgs& p, hipStream_t st);    // wave 64 x 128 (2 x 4 tiles), two workgroups per CU
int run_g1_64x256(ConvArgs& p, hipStream_t st);     // wave 32 x 128, up to three per CU
int run_g1_192x256(ConvArgs& p, hipStream_t st);    // wave 96 x 128, one per CU
int run_g1_128x256_h(ConvArgs& p, hipStream_t st);  // the same tiles on the fp16 matrix pipe (aicg_conv_desc.split == 2; conv_g1_2.hip)
int run_g1_64x256_h(ConvArgs& p, hipStream_t st);
#ifdef AICG_DEV_SWITCHES
int run_g1_256x256(ConvArgs& p, hipStream_t st);    // wave 128 x 128 (tools/kbench_g1.py)
int run_g1_128x512(ConvArgs& p, hipStream_t st);    // wave 128 x 128, four waves side by side
int run_g1_burst(ConvArgs& p, hipStream_t st, int code);   // codes 2 / 3 / 4 with the DMA of a stage as one burst
#endif

}  // namespace aicg
