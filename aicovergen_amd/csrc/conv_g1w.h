// k-tap ONE-DIMENSIONAL convolution (k = 3 / 7 / 11, unit stride, dilation 1, "same" padding, one group) in the Winograd minimal-filter
// form F(2, 3), on the machinery of conv_g1.h / conv_g1s.h: both operands by LDS DMA, single-role waves, operands picked by register NAME.
// The layers: the NSF-HiFiGAN ResBlocks of the vocoder (reference src/infer_pack/modules.py:299-312: per stage and kernel size three
// pairs c2(lrelu(c1_d(lrelu(x)))) + x; the second convolution of every pair and the first of the d = 1 pair have dilation 1 -- four of the
// six convolutions, 2/3 of the ResBlocks' 23 TFLOP per 240 s track, which conv_ws3 runs at 62-133 TFLOP/s, VERDICT r3 #4 / r4 #4: "only
// fewer multiply-adds help").
//
// Form.  Two neighbouring outputs (n, n + 1) from a 3-tap group reading d0 .. d3 = x[n + e .. n + e + 3] with FOUR products instead of six:
//     M0 += U0 (d0 - d2)   M1 += U1 (d1 + d2)   M2 += U2 (d2 - d1)   M3 += U3 (d1 - d3)        U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//     y(n) = M0 + M1 + M2      y(n + 1) = M1 - M2 - M3
// A k-tap kernel is a sum of 3-tap groups at shifted offsets e; EVERY group's products land in the same four accumulator sets (same output
// pairing, same A^T).  A remainder of two taps (a, b) takes F(2, 2): M0 += a (d0 - d1), M1 += (a + b) d1, M3 += b (d1 - d2); a single tap a:
// M0 += a d0, M3 += (-a) d1.  Products ("slots") per output pair and input channel: 4 / 10 / 15 for k = 3 / 7 / 11 where the direct form
// spends 6 / 14 / 22.  Every transform constant is +-1 or 1/2: the result differs from the direct form by fp32 rounding of the transformed
// operands and summation order (tests: 2e-6 relative against torch).
//
// Mapping.  GEMM columns are output PAIRS.  A wave owns 32 output channels x 128 outputs = two 32-pair MFMA tiles x four accumulator sets
// = 128 registers (v_mfma_f32_32x32x2_f32).  Lane l of a wave owns outputs 4 l .. 4 l + 3 (tile 0: the pair (4 l, 4 l + 1), tile 1:
// (4 l + 2, 4 l + 3)), i.e. the inputs W[i] = x[4 l - P + i], i = 0 .. k + 2 (P = (k - 1) / 2) of a channel row: value d_m of group g for
// tile j is W[2 j + 3 g + m] -- a register picked by name, the leaky ReLU of the ResBlocks and the four additions of B^T d in registers
// between the MFMAs (conv_w2d.h's division of labour: no producer waves, no transformed planes in LDS).  After the last unit the lane
// holds y of four CONSECUTIVE outputs per row: the float4 epilogue of conv_g1.h, unchanged (bias, residual, accumulate, output scale).
//
// K runs over units (channel stage, slot group): the weights of a unit are the group's <= 4 slots of the slot-major k8-interleaved image
// ([slot][K / 8][parity][Mpad][4], ops.winograd1d_kernel packed like any k-tap kernel) for 8 channels -- [slot][parity][BM] quads, a ring of
// THREE LDS buffers --, the input WINDOW of a channel stage ([channel][BN / 4 + 4 quads], from the 16-byte boundary below n0 - P) is staged
// once per stage (two buffers) and serves all its units.  k = 3 has one group: its stage is 16 channels = two units of 8, so that a
// stage's window always has a whole unit to land in.  Pipeline as conv_g1k.h: the barrier that publishes unit u + 1 sits in the middle of
// unit u, the next stage's window is issued behind the barrier of the stage's FIRST unit, the weight pieces of unit u + 2 between the
// MFMA blocks of the second half.
#pragma once
#include "conv_g1.h"

namespace aicg {

template <int K, int D = 1>
struct G1wPlan {
    static_assert(K == 3 || K == 5 || K == 7 || K == 11, "kernel sizes of the vocoder's ResBlocks (3, 7, 11) and of the flow's WaveNet layers (5)");
    static_assert(D == 1 || D == 3 || D == 5, "dilations of the vocoder's ResBlocks");
    static constexpr int P = (K - 1) / 2;
    static constexpr int NFULL = K / 3, REM = K % 3;            // 3-tap groups, remainder taps
    static constexpr int NG = NFULL + (REM ? 1 : 0);            // slot groups
    static constexpr int NSLOT = 4 * NFULL + (REM == 2 ? 3 : REM == 1 ? 2 : 0);
    static constexpr int CS = K == 3 ? 16 : 8;                  // channels per window stage
    static constexpr int NU = K == 3 ? 2 : NG;                  // units per stage
    static constexpr int BACK = (P * D + 3) / 4;                // quads the window starts below n0
    static constexpr int DELTA = 4 * BACK - P * D;              // W[i] = row[(first output of the lane) + DELTA + i D]
    // Outputs a wave owns.  D = 1: 128 (lane l: 4 l .. 4 l + 3).  D > 1: the pairs are (n, n + D), so a lane owns the progression
    // n_l + {0, D, 2 D, 3 D}, n_l = 4 D (l / D) + l % D: blocks of 4 D outputs, 30 of the 32 lanes (120 outputs) for D = 3 and 5 alike
    static constexpr int SPAN = D == 1 ? 128 : 120;
    static constexpr int LANES = D == 1 ? 32 : 30;
    __host__ __device__ static constexpr int slots_of(int g) { return g < NFULL ? 4 : (REM == 2 ? 3 : 2); }
};

// quads per window row: positions n0 - 4 BACK .. n0 + bn - 1 + P D (+ the quad a run may end in)
__host__ __device__ constexpr int g1w_row_quads(int bn, int k, int d) { return (bn + 4 * (((k - 1) / 2 * d + 3) / 4) + (k - 1) / 2 * d + 3) / 4 + 1; }

template <int I>
__device__ __forceinline__ float g1w_w(const float4 (&q)[5]) {        // W[I] out of the loaded quads (element DELTA + I of the run)
    constexpr int qi = I / 4, e = I % 4;
    static_assert(qi < 5, "window run");
    return e == 0 ? q[qi].x : e == 1 ? q[qi].y : e == 2 ? q[qi].z : q[qi].w;
}

// One scheduling group of `n` instructions of class `mask` (0x008 MFMA, 0x002 VALU, 0x100 LDS read) in the running scheduling region
#ifdef AICG_EMULATED
#define AICG_SCHED_GROUP(mask, n) ((void)0)
#else
#define AICG_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif

// SCH: the k-step regions carry an explicit interleave -- the next k-step's window reads in front of the first MFMA, its leaky ReLU and
// B^T d four VALU instructions at a time behind each of the following MFMAs -- instead of [prep][8 MFMAs] blocks (a wave issues in order:
// left as blocks, the ~25 VALU instructions of a prep sit between two MFMA bursts with this wave's share of the matrix pipe idle).
// PERS (d = 1): a workgroup WALKS tiles (blockIdx.x, + gridDim.x, ...): the unit pipeline runs on into the next tile -- its first window
// and its first two units' weights are issued under the last units of the running tile -- so that only the first tile of a workgroup
// waits for HBM in its prologue (a tile of a k = 3 layer is 50 us of work behind ~3 us of first-window latency).
// F16 (aicg_conv_desc.split == 2, the reference's is_half mode): same staging, same fp32 images, same transform B^T d in fp32; the
// operands of a unit's four k-steps are rounded to fp16 in registers -- per slot the A quad, per (tile, slot) the four k-steps' V -- and a
// unit is 2 x slots v_mfma_f32_32x32x8_f16 instead of 8 x slots fp32 MFMAs.  The next unit's operands are read, transformed and packed
// behind the running unit's MFMAs (issued, not waited for: their results are next needed by the next unit's MFMAs).
template <int K, int D, int WM, int WN, int WPS, bool PRE, int SCH = 0, bool PERS = false, bool F16 = false>
__global__ void __launch_bounds__(256) AICG_WAVES_PER_SIMD(WPS) conv_g1w_kernel(ConvArgs p) {
    using PL = G1wPlan<K, D>;
    static_assert(WM * WN == 4, "four waves");
    static_assert(!F16 || (SCH == 0 && !PERS), "the fp16 form has one schedule");
    static_assert(!PERS || D == 1, "the dilated epilogue's LDS tiles overlay the pipeline's buffers");
    constexpr int BM = 32 * WM, BN = PL::SPAN * WN;
    constexpr int RQ = g1w_row_quads(BN, K, D);
    constexpr int ASTAGE = 4 * 2 * BM * 4;               // floats: <= 4 slots x 2 parities x BM quads
    constexpr int BQ = PL::CS * RQ, NB = (BQ + 63) / 64, BSTAGE = NB * 256;
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* const smem = reinterpret_cast<float*>(smem4);
    float* const bbuf = smem + 3 * ASTAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    const int nwg_all = p.N * p.tiles_h * p.tiles_w;     // tiles of the launch (PERS: more than workgroups)
    const int nst = (p.Cin_g + PL::CS - 1) / PL::CS;
    // tile-dependent state: the running tile's and (PERS) the next one's -- coordinates, the image's base, this wave's DMA offsets
    int m_base, n0, img, s0;
    const float* ximg;
    auto coords = [&](int hw) __attribute__((always_inline)) {
        const int bid = (int)xcd_remap((unsigned)hw, (unsigned)nwg_all);
        const int mt = bid % p.tiles_h;
        const int ct = (bid / p.tiles_h) % p.tiles_w;
        img = bid / (p.tiles_h * p.tiles_w);
        m_base = mt * BM;
        n0 = ct * BN;
        s0 = n0 - 4 * PL::BACK;                          // first position of the window
        ximg = p.x + (long)img * p.x_sn;
    };
    coords((int)blockIdx.x);

    // ---- DMA plan.  Weights of unit (stage cs, v): slots sl0 .. sl0 + ns - 1 of the 8-channel block kb = cs CS / 8 + (K == 3 ? v : 0)
    const long wslot_q = (long)(p.Cin_pad >> 3) * 2 * p.Mpad;              // quads of one slot's image
    const BufRsrc wb = make_buf(p.w3, (unsigned)lmin((long)PL::NSLOT * wslot_q * 16, 0x7fffffffL));
    // per-lane byte offsets of this wave's pieces, computed ONCE per tile (a piece inside the unit pipeline is then M0 + one buffer
    // instruction): weights -- quad q = piece 64 + lane of [slot][parity][BM] -- relative to the unit's (first slot, 8-channel block); the
    // window -- quad q of [channel][RQ] -- relative to the stage's first channel
    constexpr int PA = (4 * 2 * BM / 64 + 3) / 4, PB = (NB + 3) / 4;
    unsigned aoff[PA], boff[PB];                          // the running tile's
    unsigned aoff1[PERS ? PA : 1], boff1[PERS ? PB : 1];  // PERS: the next tile's
    const float* ximg1 = nullptr;
    bool has1 = false;
    auto offsets = [&](unsigned (&ao)[PA], unsigned (&bo)[PB]) __attribute__((always_inline)) {   // from m_base, s0
#pragma unroll
        for (int e = 0; e < PA; ++e) {
            const int q = (wave + 4 * e) * 64 + lane;
            const int sl = q / (2 * BM), rem = q - sl * 2 * BM;
            const int par = rem / BM, m = rem - par * BM;
            ao[e] = m_base + m < p.Mpad ? 16u * (unsigned)(sl * wslot_q + (long)par * p.Mpad + m_base + m) : kBufOob;
        }
#pragma unroll
        for (int e = 0; e < PB; ++e) {
            const int q = (wave + 4 * e) * 64 + lane;
            const int row = q / RQ, col = q - row * RQ;
            const int pos = s0 + 4 * col;
            bo[e] = (q < BQ && pos >= 0 && pos < p.W) ? 4u * (unsigned)(row * (int)p.x_sc + pos) : kBufOob;   // (W % 4 == 0: whole quads)
        }
    };
    offsets(aoff, boff);
    auto issue_a = [&](const unsigned (&ao)[PA], int kb, int sl0, int ns, float* abuf) __attribute__((always_inline)) {
        const int npiece = ns * 2 * BM / 64;              // [slot][parity][BM] quads in 64-quad pieces
        const unsigned soff = (unsigned)(((long)sl0 * wslot_q + (long)kb * 2 * p.Mpad) * 16);
#pragma unroll
        for (int e = 0; e < PA; ++e)
            if (wave + 4 * e < npiece) w2d_dma16(wb, ao[e], soff, abuf + (wave + 4 * e) * 256, lane);
    };
    auto issue_b = [&](const unsigned (&bo)[PB], const float* xi, int cs, float* dst) __attribute__((always_inline)) {
        const long left = (long)(p.Cin_g - cs * PL::CS) * p.x_sc * 4;       // absent channels read 0
        const BufRsrc xb = make_buf(xi + (long)cs * PL::CS * p.x_sc, (unsigned)lmin(left, 0x7fffffffL));
#pragma unroll
        for (int e = 0; e < PB; ++e)
            if (4 * e + 3 < NB || wave + 4 * e < NB) w2d_dma16(xb, bo[e], 0u, dst + (wave + 4 * e) * 256, lane);
    };

    f32x16 M[4][2];                                       // [accumulator set][pair tile]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) M[q][j][r] = 0.f;

    const int a_lane = half * BM + wm * 32 + l31;          // float4 index inside a slot's slab pair
    const int b_lane = half * RQ + wn * 32 + l31;          // D = 1: float4 index of the lane's first window quad inside a row pair
    // D > 1: first output of the lane inside the wave's span, and the float index of its W[0] inside a row pair
    const int la = l31 < PL::LANES ? l31 : PL::LANES - 1;  // (lanes 30, 31 shadow lane 29; their results are not stored)
    const int n_lane = 4 * D * (la / D) + la % D;
    const int f_lane = half * RQ * 4 + wn * PL::SPAN + n_lane + PL::DELTA;
    const float pre_slope = p.pre_slope;
    // lrelu(v) = max(v, slope v) for 0 <= slope <= 1: a multiply and ONE v_max_f32 (inline asm: fmaxf() -- and fmed3(a, b, inf), which LLVM
    // folds back into maxnum -- adds a canonicalising v_max(x, x) per value; same bits for every non-NaN v)
    auto lrelu = [&](float v) __attribute__((always_inline)) {
        if constexpr (!PRE) return v;
#ifdef AICG_EMULATED
        return fmaxf(v, v * pre_slope);
#else
        const float sv = v * pre_slope;
        float r;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(sv));
        return r;
#endif
    };

    // ---- the k-step pipeline.  A k-step of unit (cs, v) -- group G, channels rb + 2 s + half of the stage's window -- has two parts:
    //   prep: the window quads the group's two tiles need (<= 3 ds_read_b128), the leaky ReLU and B^T d in registers -> V[tile][slot]
    //   mma : 2 x slots MFMAs with the unit's A fragments (one float4 per slot = its four k-steps)
    // and prep of k-step i + 1 is issued in front of the MFMAs of k-step i (across unit and stage boundaries), so that LDS latency and
    // the transform's VALU sit under 512 cycles of matrix pipe instead of in front of them.
    auto prep = [&](auto g_tag, int s, const float* wbuf, int rb, float (&V)[2][4]) __attribute__((always_inline)) {
        constexpr int G = decltype(g_tag)::value;
        constexpr int NS = PL::slots_of(G);
        constexpr int I0 = PL::DELTA + 3 * G;              // element of the quad run that is d0 of tile 0
        constexpr int Q0 = I0 / 4, Q1 = (I0 + 2 + 3) / 4;  // quads the two tiles' d0 .. d3 live in (tile 1 is two positions on)
        constexpr int NQ = Q1 - Q0 + 1;
        constexpr int base = I0 - 4 * Q0;
        static_assert(NQ <= 3, "two tiles of a group span at most three quads");
        float w0, w1, w2, w3, w4, w5;
        if constexpr (D == 1) {
            float4 q[5];
            const float4* xt = reinterpret_cast<const float4*>(__builtin_assume_aligned(wbuf, 16)) + b_lane + (rb + 2 * s) * RQ + Q0;
#pragma unroll
            for (int t = 0; t < NQ; ++t) q[t] = xt[t];
            w0 = lrelu(g1w_w<base + 0>(q)); w1 = lrelu(g1w_w<base + 1>(q)); w2 = lrelu(g1w_w<base + 2>(q));
            w3 = lrelu(g1w_w<base + 3>(q)); w4 = lrelu(g1w_w<base + 4>(q)); w5 = lrelu(g1w_w<base + 5>(q));
        } else {
            // the six values W[3 G + e], e = 0 .. 5, sit D positions apart: six ds_read_b32 at immediate offsets from the lane's base
            const float* xr = wbuf + f_lane + (rb + 2 * s) * RQ * 4 + 3 * G * D;
            w0 = lrelu(xr[0]); w1 = lrelu(xr[D]); w2 = lrelu(xr[2 * D]); w3 = lrelu(xr[3 * D]); w4 = lrelu(xr[4 * D]); w5 = lrelu(xr[5 * D]);
        }
        const float d[2][4] = {{w0, w1, w2, w3}, {w2, w3, w4, w5}};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (NS == 4) {
                V[j][0] = d[j][0] - d[j][2]; V[j][1] = d[j][1] + d[j][2]; V[j][2] = d[j][2] - d[j][1]; V[j][3] = d[j][1] - d[j][3];
            } else if constexpr (NS == 3) {                // two taps (a, b): slots a -> M0, a + b -> M1, b -> M3
                V[j][0] = d[j][0] - d[j][1]; V[j][1] = d[j][1]; V[j][2] = d[j][1] - d[j][2]; V[j][3] = 0.f;
            } else {                                       // one tap a: slots a -> M0, -a -> M3
                V[j][0] = d[j][0]; V[j][1] = d[j][1]; V[j][2] = 0.f; V[j][3] = 0.f;
            }
        }
    };
    auto load_a = [&](auto g_tag, const float* abuf, float4 (&a)[4]) __attribute__((always_inline)) {
        constexpr int NS = PL::slots_of(decltype(g_tag)::value);
        const float4* wt = reinterpret_cast<const float4*>(__builtin_assume_aligned(abuf, 16)) + a_lane;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) a[sl] = wt[sl * 2 * BM];
    };
    auto mma = [&](auto g_tag, int s, const float4 (&a)[4], const float (&V)[2][4]) __attribute__((always_inline)) {
        constexpr int NS = PL::slots_of(decltype(g_tag)::value);
        auto av = [&](int sl) __attribute__((always_inline)) { return s == 0 ? a[sl].x : s == 1 ? a[sl].y : s == 2 ? a[sl].z : a[sl].w; };
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (NS == 4) {
                M[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(0), V[j][0], M[0][j], 0, 0, 0);
                M[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(1), V[j][1], M[1][j], 0, 0, 0);
                M[2][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(2), V[j][2], M[2][j], 0, 0, 0);
                M[3][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(3), V[j][3], M[3][j], 0, 0, 0);
            } else if constexpr (NS == 3) {
                M[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(0), V[j][0], M[0][j], 0, 0, 0);
                M[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(1), V[j][1], M[1][j], 0, 0, 0);
                M[3][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(2), V[j][2], M[3][j], 0, 0, 0);
            } else {
                M[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(0), V[j][0], M[0][j], 0, 0, 0);
                M[3][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av(1), V[j][1], M[3][j], 0, 0, 0);
            }
        }
    };

    // ---- the unit pipeline.  Unit index u = cs NU + v; weights ring of three (rotating per unit, across tiles), windows bbuf[wpar] (the
    // running stage's) and bbuf[wpar ^ 1] (the next stage's -- the next TILE's first one behind a tile's last stage when PERS).
    const int nunits = nst * PL::NU;
    float* a_cur = smem;
    float* a_nxt = smem + ASTAGE;
    float* a_fill = smem + 2 * ASTAGE;
    int wpar = 0;
    auto issue_unit = [&](const unsigned (&ao)[PA], int cs, int v, float* abuf) __attribute__((always_inline)) {
        if constexpr (K == 3) issue_a(ao, cs * 2 + v, 0, 4, abuf);
        else issue_a(ao, cs, 4 * v, v < PL::NFULL ? 4 : PL::slots_of(PL::NFULL), abuf);
    };
    issue_b(boff, ximg, 0, bbuf);
    issue_unit(aoff, 0, 0, a_cur);
    issue_unit(aoff, 1 / PL::NU, 1 % PL::NU, a_nxt);      // (NU >= 2: unit 1 exists)
    g1_wait_pieces<0>();
    lds_barrier();
    float4 af[4], an[4];                                  // A fragments of the running unit / of the next one
    float Vc[2][4], Vn[2][4];                             // operands of the running k-step / of the next one
    auto open_tile = [&]() __attribute__((always_inline)) {   // operands of a tile's first k-step, with nothing to hide behind
        load_a(std::integral_constant<int, 0>{}, a_cur, af);
        prep(std::integral_constant<int, 0>{}, 0, bbuf + wpar * BSTAGE, 0, Vc);
    };
    if constexpr (!F16) open_tile();
    // One unit of the walk; V (unit inside the stage) at compile time.  MODE 0: a unit with a successor in the same tile (the next unit's
    // operands are prefetched into registers under this one's last k-steps); 1: a tile's last unit with another tile behind it (PERS: the
    // barrier and the DMA issue go on, the register prefetch does not -- the epilogue in between needs the registers); 2: the walk's last.
    auto run_unit_f = [&](auto v_tag, auto mode_tag, int cs) __attribute__((always_inline)) {
        constexpr int V = decltype(v_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool PREF = MODE == 0, SYNC = MODE != 2;
        constexpr int G = K == 3 ? 0 : V;
        constexpr int VN = (V + 1) % PL::NU, GN = K == 3 ? 0 : VN;      // the unit behind this one
        using GT = std::integral_constant<int, G>;
        using GNT = std::integral_constant<int, GN>;
        const int rb = K == 3 ? 8 * V : 0, rbn = K == 3 ? 8 * VN : 0;
        const float* wbuf = bbuf + wpar * BSTAGE;
        const float* wnxt = bbuf + (V + 1 == PL::NU ? wpar ^ 1 : wpar) * BSTAGE;
        // (a k-step region: prep of the next k-step + the MFMAs of this one, interleaved when SCH)
        auto interleave = [&]() __attribute__((always_inline)) {
            if constexpr (SCH != 0) {
                AICG_SCHED_GROUP(0x100, 3);
                AICG_SCHED_GROUP(0x008, 1);
                AICG_SCHED_GROUP(0x002, 2);
                AICG_SCHED_GROUP(0x008, 1);
#pragma unroll
                for (int e = 0; e < 6; ++e) { AICG_SCHED_GROUP(0x002, 4); AICG_SCHED_GROUP(0x008, 1); }
                AICG_SCHED_GROUP(0x002, 8);
            }
        };
        // k-steps 0 and 1
        w2d_fence();
        prep(GT{}, 1, wbuf, rb, Vn);
        if constexpr (SCH == 0) w2d_fence();
        mma(GT{}, 0, af, Vc);
        interleave();
        w2d_fence();
        prep(GT{}, 2, wbuf, rb, Vc);
        if constexpr (SCH == 0) w2d_fence();
        mma(GT{}, 1, af, Vn);
        interleave();
        w2d_fence();
        if constexpr (SYNC) {
            g1_wait_pieces<0>();   // this wave's pieces of unit u + 1 (and, issued in front of them, the next stage's window)
            lds_barrier();         // unit u + 1 (and, behind a stage's last unit, the next window) is complete; unit u - 1's buffer is free
            if constexpr (V == 0) {
                if (cs + 1 < nst) issue_b(boff, ximg, cs + 1, bbuf + (wpar ^ 1) * BSTAGE);
                else if constexpr (PERS) { if (has1) issue_b(boff1, ximg1, 0, bbuf + (wpar ^ 1) * BSTAGE); }
            }
            {   // unit u + 2 = (cs + (V + 2) / NU, (V + 2) % NU): known at compile time up to cs; behind the tile's end: the next tile's
                constexpr int V2 = (V + 2) % PL::NU;
                const int cs2 = cs + (V + 2) / PL::NU;
                if (cs2 < nst) issue_unit(aoff, cs2, V2, a_fill);
                else if constexpr (PERS) { if (has1) issue_unit(aoff1, cs2 - nst, V2, a_fill); }
            }
            if constexpr (PREF) load_a(GNT{}, a_nxt, an);
        }
        // k-steps 2 and 3
        w2d_fence();
        prep(GT{}, 3, wbuf, rb, Vn);
        if constexpr (SCH == 0) w2d_fence();
        mma(GT{}, 2, af, Vc);
        interleave();
        w2d_fence();
        if constexpr (PREF) prep(GNT{}, 0, wnxt, rbn, Vc);
        if constexpr (SCH == 0) w2d_fence();
        mma(GT{}, 3, af, Vn);
        if constexpr (PREF) interleave();
        w2d_fence();
        if constexpr (PREF) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) af[sl] = an[sl];
        }
        float* t = a_cur; a_cur = a_nxt; a_nxt = a_fill; a_fill = t;
        if constexpr (V + 1 == PL::NU) wpar ^= 1;
    };
    // ---- F16: a unit's operands packed in registers
    H4 ah[4], Vh[2][4];                                   // running unit: A per slot, V per (tile, slot): the four k-steps' values
    auto pack_unit = [&](auto g_tag, const float* abuf, const float* wsrc, int rb_) __attribute__((always_inline)) {
        constexpr int NS = PL::slots_of(decltype(g_tag)::value);
        float4 a4[4];
        load_a(g_tag, abuf, a4);
        float Vf[4][2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) prep(g_tag, s, wsrc, rb_, Vf[s]);
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) ah[sl] = pack_f16x4(a4[sl].x, a4[sl].y, a4[sl].z, a4[sl].w);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) Vh[j][sl] = pack_f16x4(Vf[0][j][sl], Vf[1][j][sl], Vf[2][j][sl], Vf[3][j][sl]);
    };
    auto run_unit_h = [&](auto v_tag, auto mode_tag, int cs) __attribute__((always_inline)) {
        constexpr int V = decltype(v_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool PREF = MODE == 0, SYNC = MODE != 2;
        constexpr int G = K == 3 ? 0 : V;
        constexpr int VN = (V + 1) % PL::NU, GN = K == 3 ? 0 : VN;
        constexpr int NS = PL::slots_of(G);
        using GNT = std::integral_constant<int, GN>;
        const int rbn = K == 3 ? 8 * VN : 0;
        const float* wnxt = bbuf + (V + 1 == PL::NU ? wpar ^ 1 : wpar) * BSTAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (NS == 4) {
                M[0][j] = mfma_f16_32x32x8(ah[0], Vh[j][0], M[0][j]); M[1][j] = mfma_f16_32x32x8(ah[1], Vh[j][1], M[1][j]);
                M[2][j] = mfma_f16_32x32x8(ah[2], Vh[j][2], M[2][j]); M[3][j] = mfma_f16_32x32x8(ah[3], Vh[j][3], M[3][j]);
            } else if constexpr (NS == 3) {
                M[0][j] = mfma_f16_32x32x8(ah[0], Vh[j][0], M[0][j]); M[1][j] = mfma_f16_32x32x8(ah[1], Vh[j][1], M[1][j]);
                M[3][j] = mfma_f16_32x32x8(ah[2], Vh[j][2], M[3][j]);
            } else {
                M[0][j] = mfma_f16_32x32x8(ah[0], Vh[j][0], M[0][j]); M[3][j] = mfma_f16_32x32x8(ah[1], Vh[j][1], M[3][j]);
            }
        }
        w2d_fence();
        if constexpr (SYNC) {
            g1_wait_pieces<0>();
            lds_barrier();
            if constexpr (V == 0) {
                if (cs + 1 < nst) issue_b(boff, ximg, cs + 1, bbuf + (wpar ^ 1) * BSTAGE);
            }
            {
                constexpr int V2 = (V + 2) % PL::NU;
                const int cs2 = cs + (V + 2) / PL::NU;
                if (cs2 < nst) issue_unit(aoff, cs2, V2, a_fill);
            }
            if constexpr (PREF) pack_unit(GNT{}, a_nxt, wnxt, rbn);
        }
        w2d_fence();
        float* t = a_cur; a_cur = a_nxt; a_nxt = a_fill; a_fill = t;
        if constexpr (V + 1 == PL::NU) wpar ^= 1;
    };
    using M0 = std::integral_constant<int, 0>;
    // the units of one tile; END = the mode of its last unit
    auto run_tile = [&](auto end_tag) __attribute__((always_inline)) {
        using END = decltype(end_tag);
        auto run_unit = [&](auto v_tag, auto mode_tag, int cs_) __attribute__((always_inline)) {
            if constexpr (F16) run_unit_h(v_tag, mode_tag, cs_);
            else run_unit_f(v_tag, mode_tag, cs_);
        };
        for (int cs = 0; cs + 1 < nst; ++cs) {
            run_unit(std::integral_constant<int, 0>{}, M0{}, cs);
            if constexpr (PL::NU > 1) run_unit(std::integral_constant<int, 1>{}, M0{}, cs);
            if constexpr (PL::NU > 2) run_unit(std::integral_constant<int, 2>{}, M0{}, cs);
            if constexpr (PL::NU > 3) run_unit(std::integral_constant<int, 3>{}, M0{}, cs);
        }
        const int cs = nst - 1;   // the last stage, peeled: its last unit ends the tile
        if constexpr (PL::NU == 2) { run_unit(std::integral_constant<int, 0>{}, M0{}, cs); run_unit(std::integral_constant<int, 1>{}, END{}, cs); }
        if constexpr (PL::NU == 3) {
            run_unit(std::integral_constant<int, 0>{}, M0{}, cs); run_unit(std::integral_constant<int, 1>{}, M0{}, cs);
            run_unit(std::integral_constant<int, 2>{}, END{}, cs);
        }
        if constexpr (PL::NU == 4) {
            run_unit(std::integral_constant<int, 0>{}, M0{}, cs); run_unit(std::integral_constant<int, 1>{}, M0{}, cs);
            run_unit(std::integral_constant<int, 2>{}, M0{}, cs); run_unit(std::integral_constant<int, 3>{}, END{}, cs);
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
    // ---- A^T: the lane's four outputs per row
    f32x16 y[1][4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        y[0][0][r] = (M[0][0][r] + M[1][0][r]) + M[2][0][r];
        y[0][1][r] = (M[1][0][r] - M[2][0][r]) - M[3][0][r];
        y[0][2][r] = (M[0][1][r] + M[1][1][r]) + M[2][1][r];
        y[0][3][r] = (M[1][1][r] - M[2][1][r]) - M[3][1][r];
    }
    if constexpr (D == 1) {
        // four CONSECUTIVE outputs: conv_g1.h's float4 epilogue
        g1_epilogue<1, false, false>(p, y, img, m_base + wm * 32, n0 + wn * 128 + 4 * l31, p.Wo);
    } else {
        // the outputs n_l + {0, D, 2 D, 3 D}: through a per-wave LDS tile [32 rows][SPAN + 4] (the pipeline's buffers are free: every wave
        // is past the last unit's barrier ... after one more), read back as float4 runs of a row -- 15 per lane -- and stored with the
        // bias / activation / residual / accumulate arithmetic of conv_g1.h's epilogue
        constexpr int LD = PL::SPAN + 4;
        lds_barrier();
        float* tile = smem + wave * 32 * LD;
        if (l31 < PL::LANES) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* row = tile + (8 * (r >> 2) + 4 * half + (r & 3)) * LD + n_lane;
                row[0] = y[0][0][r]; row[D] = y[0][1][r]; row[2 * D] = y[0][2][r]; row[3 * D] = y[0][3][r];
            }
        }
        __builtin_amdgcn_wave_barrier();   // (a wave's LDS accesses are processed in order; the emulator's lanes rendezvous here)
        const int m0 = m_base + wm * 32, nw = n0 + wn * PL::SPAN;
        auto body = [&](auto act_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
            for (int idx = lane; idx < 32 * (PL::SPAN / 4); idx += 64) {
                const int rw = idx / (PL::SPAN / 4), q4 = idx - rw * (PL::SPAN / 4);
                const int m = m0 + rw, pos = nw + 4 * q4;
                if (m >= p.Cout_g || pos >= p.Wo) continue;          // (Wo % 4 == 0: a quad is inside the row or outside)
                const float4 v4 = *reinterpret_cast<const float4*>(tile + rw * LD + 4 * q4);
                const float bv = p.bias ? p.bias[m] : 0.f;
                float v[4] = {v4.x + bv, v4.y + bv, v4.z + bv, v4.w + bv};
                float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), yv = make_float4(0.f, 0.f, 0.f, 0.f);
                float* dst = p.y + (long)img * p.y_sn + (long)m * p.y_sc + pos;
                if (p.res) rv = *reinterpret_cast<const float4*>(p.res + (long)img * p.r_sn + (long)m * p.r_sc + pos);
                if (p.accumulate) yv = *reinterpret_cast<const float4*>(dst);
                const float rr[4] = {rv.x, rv.y, rv.z, rv.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float x = v[t];
                    if (p.res_first) x += rr[t];
                    x = act_static<ACT>(x, p.act, p.act_slope);
                    if (!p.res_first) x += rr[t];
                    v[t] = x * p.out_scale + yy[t];
                }
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            }
        };
        if (p.act == AICG_ACT_NONE) body(std::integral_constant<int, 0>{});
        else if (p.act == AICG_ACT_RELU) body(std::integral_constant<int, 1>{});
        else if (p.act == AICG_ACT_LRELU) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 3>{});
    }
    };
    if constexpr (!PERS) {
        if constexpr (F16) pack_unit(std::integral_constant<int, 0>{}, a_cur, bbuf + wpar * BSTAGE, 0);
        run_tile(std::integral_constant<int, 2>{});
        epilogue();
    } else {
        // the walk: tile k of this workgroup is tile blockIdx.x + k gridDim.x of the launch; `has1` = the tile behind the running one
        int hw = (int)blockIdx.x + (int)gridDim.x;
        auto place_next = [&]() __attribute__((always_inline)) {
            has1 = hw < nwg_all;
            if (has1) {
                const int mb = m_base, nn = n0, im = img, ss = s0;
                const float* xi = ximg;
                coords(hw);
                offsets(aoff1, boff1);
                ximg1 = ximg;
                m_base = mb; n0 = nn; img = im; s0 = ss; ximg = xi;
            }
            hw += (int)gridDim.x;
        };
        place_next();
        while (true) {
            if (has1) run_tile(std::integral_constant<int, 1>{});
            else run_tile(std::integral_constant<int, 2>{});
            epilogue();
            if (!has1) break;
            // the next tile becomes the running one: its first window and its units 0 / 1 are in flight or landed already
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) M[q][j][r] = 0.f;
            coords(hw - (int)gridDim.x);
#pragma unroll
            for (int e = 0; e < PA; ++e) aoff[e] = aoff1[e];
#pragma unroll
            for (int e = 0; e < PB; ++e) boff[e] = boff1[e];
            place_next();
            open_tile();
        }
    }
}

// host side: a 1-D layer of the form this kernel takes
inline bool conv_g1w_applicable(const ConvArgs& p, int pad_w_end) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    auto m4 = [](long v) { return (v & 3) == 0; };
    if (p.KH != 1 || p.H != 1 || p.Ho != 1 || (p.KW != 3 && p.KW != 5 && p.KW != 7 && p.KW != 11) || p.groups != 1 || p.sw != 1 || !p.w3) return false;
    if (p.KW == 5 && p.dw != 1) return false;
    if (p.dw != 1 && p.dw != 3 && p.dw != 5) return false;
    if (p.pw != (p.KW - 1) / 2 * p.dw || pad_w_end != p.pw || p.ph || p.Wo != p.W || (p.W & 3) || p.Cin_g < 16) return false;
    if (p.pre_act != AICG_ACT_NONE && !(p.pre_act == AICG_ACT_LRELU && p.pre_slope >= 0.f && p.pre_slope <= 1.f)) return false;
    if (p.shuffle || p.res_mul || p.W >= (1 << 24) || p.x_sc >= (1L << 24) || p.x_sc < p.W) return false;
    if (!al(p.x) || !m4(p.x_sn) || !m4(p.x_sc) || !al(p.y) || !m4(p.y_sn) || !m4(p.y_sc)) return false;
    if (p.res && (!al(p.res) || !m4(p.r_sn) || !m4(p.r_sc))) return false;
    const long nslot = p.KW == 3 ? 4 : p.KW == 5 ? 7 : p.KW == 7 ? 10 : 15;
    if (nslot * p.Cin_pad * p.Mpad * 4 >= (1L << 31)) return false;                        // 32-bit byte offsets inside the slot image
    return true;
}

template <int K, int D, int WM, int WN, int WPS, int SCH, bool PERS = false, bool F16 = false>
static int launch_conv_g1w_kd(ConvArgs& p, hipStream_t stream) {
    using PL = G1wPlan<K, D>;
    constexpr int BM = 32 * WM, BN = PL::SPAN * WN;
    p.tiles_h = idiv_up(p.Cout_g, BM);
    p.tiles_w = idiv_up(p.Wo, BN);
    const long nwg = (long)p.N * p.tiles_h * p.tiles_w;
    if (nwg > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    size_t lds = (size_t)(3 * 4 * 2 * BM * 4 + 2 * ((PL::CS * g1w_row_quads(BN, K, D) + 63) / 64) * 256) * sizeof(float);
    if (D > 1 && lds < (size_t)4 * 32 * (PL::SPAN + 4) * sizeof(float)) lds = (size_t)4 * 32 * (PL::SPAN + 4) * sizeof(float);   // the epilogue's tiles
    if (lds > 160 * 1024) return 1;
    auto kern = p.pre_act != AICG_ACT_NONE ? conv_g1w_kernel<K, D, WM, WN, WPS, true, SCH, PERS, F16> : conv_g1w_kernel<K, D, WM, WN, WPS, false, SCH, PERS, F16>;
    allow_dynamic_lds((const void*)kern, lds);
    long grid = nwg;
    const long slots = p.dbg > 0 ? p.dbg : 2 * 256;      // two workgroups per CU, each walking tiles blockIdx.x, + grid, ... (p.dbg: tests)
    if (PERS && nwg > slots) grid = slots;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, p);
    return check_launch("conv_g1w_kernel");
}

template <int WM, int WN, int WPS, int SCH = 0, bool DIL = true, bool PERS = false, bool F16 = false>
static int launch_conv_g1w(ConvArgs& p, hipStream_t stream) {
    if (p.dw == 1) {
        if (p.KW == 3) return launch_conv_g1w_kd<3, 1, WM, WN, WPS, SCH, PERS, F16>(p, stream);
        if (p.KW == 5) return launch_conv_g1w_kd<5, 1, WM, WN, WPS, SCH, PERS, F16>(p, stream);
        if (p.KW == 7) return launch_conv_g1w_kd<7, 1, WM, WN, WPS, SCH, PERS, F16>(p, stream);
        return launch_conv_g1w_kd<11, 1, WM, WN, WPS, SCH, PERS, F16>(p, stream);
    }
    if constexpr (DIL) {
        if (p.dw == 3) {
            if (p.KW == 3) return launch_conv_g1w_kd<3, 3, WM, WN, WPS, SCH, false, F16>(p, stream);
            if (p.KW == 7) return launch_conv_g1w_kd<7, 3, WM, WN, WPS, SCH, false, F16>(p, stream);
            return launch_conv_g1w_kd<11, 3, WM, WN, WPS, SCH, false, F16>(p, stream);
        }
        if (p.KW == 3) return launch_conv_g1w_kd<3, 5, WM, WN, WPS, SCH, false, F16>(p, stream);
        if (p.KW == 7) return launch_conv_g1w_kd<7, 5, WM, WN, WPS, SCH, false, F16>(p, stream);
        return launch_conv_g1w_kd<11, 5, WM, WN, WPS, SCH, false, F16>(p, stream);
    }
    return 1;
}

// instantiation unit conv_g1w_1.hip
int run_g1w_64x256(ConvArgs& p, hipStream_t st);    // 2 x 2 waves of 32 rows x 128 outputs
int run_g1w_32x512(ConvArgs& p, hipStream_t st);    // 1 x 4 waves: all four share the tile's 32 rows
int run_g1w_32x512_h(ConvArgs& p, hipStream_t st);  // ... on the fp16 matrix pipe (aicg_conv_desc.split == 2; conv_g1w_4.hip)
int run_g1w_32x512_sched(ConvArgs& p, hipStream_t st);   // ... with the explicit MFMA / VALU interleave (SCH)
int run_g1w_32x512_pers(ConvArgs& p, hipStream_t st);    // ... as a persistent tile walk (d = 1)

}  // namespace aicg
