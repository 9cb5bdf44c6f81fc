// 3 x 3 convolution (stride 1, dilation 1, padding 1) by the Winograd minimal-filtering form F(2, 3) ALONG THE ROW AXIS:
// two neighbouring outputs of a row from four inputs with 4 multiplications per (channel, kernel row) instead of 6 --
// 12 MFMA contractions per output pair, input channel and output channel where the direct implicit GEMM spends 18.
//
//     Y = A^T [ sum_{ci, kh} (G g[ci][kh][0..2]) (.) (B^T d[ci][row + kh][2 j - 1 .. 2 j + 2]) ]
//     B^T d: V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
//     G g  : U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2          (folded into the packed weights at load time)
//     A^T M: Y(2 j) = M0 + M1 + M2,  Y(2 j + 1) = M1 - M2 - M3                            (the bias is the initial value of M1)
//
// so the layer becomes FOUR implicit GEMMs (one per Winograd point p) of K = 3 Cin over half as many columns: the weights are packed
// as a (Cout, Cin, 3, 4) kernel whose "column taps" are the points -- the k8-interleaved image of conv_ws3.h, unchanged -- the
// producer waves build the four V planes of an 8-channel chunk while they stage it, and a consumer wave keeps four accumulator
// sets, one per point, that meet only in the epilogue.  Serves the MDX-Net TFC convolutions (a2: 3 x 3 + BatchNorm + ReLU, Cout a
// multiple of 32): the largest block of the convolution family's time.  fp32 throughout; the transform constants are exact (+-1,
// 1/2), the result differs from the direct form only by fp32 summation order (tests: 2e-6 against torch).
//
// Geometry: a workgroup (ROWS consumer + 4 producer waves, one workgroup per CU) owns BM output channels x ROWS output rows x 64
// output columns; consumer wave w owns row w and all four points of it, 32 column pairs wide.
//   BM = 32 TM (32 / 64 / 96), ROWS = 4: v_mfma_f32_32x32x2_f32, lane j = pair j, 64 TM accumulator registers per point (192 of the
//       256 a wave has with two waves per SIMD at TM = 3).
//   BM = 48, ROWS = 8 (MDX-Net's 48- and 144-channel levels): v_mfma_f32_16x16x4_f32, 3 row blocks x 2 pair blocks, 96 accumulator
//       registers -- TWO consumer waves per SIMD, so one wave's fragment hand-over bubbles are the other's MFMA time; a lane reads
//       the 8-byte half of the same quads that holds its k slot's channels (slot s = lane >> 4: parity s & 1, elements 2 (s >> 1) + {0, 1}).
// LDS: patch [chunk buffer][parity][point][input row 0..ROWS+1][pair j][4 channels], weights [buffer][tap = kh * 4 + p][parity][BM][4],
// the bias slice.  One stage = one 8-channel chunk = 12 taps.
#pragma once
#include "conv_kernels.h"

namespace aicg {

static constexpr int kWinoPairs = 32;   // column pairs per workgroup

// A per-tile copy of a uniform value the optimiser cannot see through: the epilogue's channel offsets are the same for every tile of
// the walk, and hoisted out of the tile loop they would occupy (and spill) the registers the accumulators need.
template <class T>
__device__ inline T per_tile(T v) {
#ifndef AICG_EMULATED
    asm volatile("" : "+s"(v));
#endif
    return v;
}
__device__ inline int per_tile_lane(int v) {   // the same for a per-lane value
#ifndef AICG_EMULATED
    asm volatile("" : "+v"(v));
#endif
    return v;
}
__device__ __forceinline__ void sched_fence() {   // nothing moves across
#ifndef AICG_EMULATED
    __builtin_amdgcn_sched_barrier(0);
#endif
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int BM, int ROWS>
__global__ void __launch_bounds__(64 * ROWS + 256) conv_ws3w_kernel(ConvArgs p) {
    constexpr bool M16 = BM == 48;                   // 16 x 16 x 4 MFMAs (else 32 x 32 x 2)
    constexpr int TM = BM / 32;                      // row blocks of the 32 x 32 form
    constexpr int NC = 64 * ROWS;                    // consumer threads
    constexpr int PROWS = ROWS + 2;                  // patch rows
    constexpr int PLANE = PROWS * kWinoPairs;        // float4 per (parity, point) plane
    constexpr int XS4 = 2 * 4 * PLANE;               // float4 per patch buffer
    constexpr int WS4 = 24 * BM;                     // float4 per weight stage: 12 taps x 2 parities x BM
    constexpr int WR = (WS4 + 255) / 256;            // float4 weight copies per producer thread and stage
    constexpr int NI = (2 * PLANE + 255) / 256;      // patch items (parity, input row, pair) per producer thread and stage
    HIP_DYNAMIC_SHARED(float4, smem4)
    float4* const xs0 = smem4;                       // 2 x XS4
    float4* const ws0 = smem4 + 2 * XS4;             // 2 x WS4
    float* const bias_s = reinterpret_cast<float*>(smem4 + 2 * XS4 + 2 * WS4);   // BM: this workgroup's slice of the bias
    const int tid = threadIdx.x;
    // Persistent tile walk: the launch has 8 x `slots` workgroups per output-channel tile (one per CU); the workgroups of XCD x share
    // that XCD's contiguous eighth of the tile list and take it `slots` at a time, so the rows a tile shares with its neighbours
    // are in the same L2 at about the same time.  The producers run ahead across tile boundaries: the first chunks of the next tile
    // are staged under this tile's last chunk and epilogue, and the epilogue's stores drain under the next tile's MFMAs.
    const int ntiles = p.N * p.tiles_h * p.tiles_w;
    const int slots = gridDim.x >> 3, per_xcd = (ntiles + 7) >> 3;
    const int first = (blockIdx.x & 7) * per_xcd, slot = blockIdx.x >> 3;
    const int mine = ntiles - first < per_xcd ? ntiles - first : per_xcd;            // tiles in this XCD's share
    const int my_tiles = slot < mine ? (mine - slot + slots - 1) / slots : 0;        // tile k of this workgroup: first + slot + k * slots
    const int m_base = blockIdx.y * BM;
    const int nchunk = p.nchunk;                     // 8-channel chunks

    // The bias seeds the accumulators of point 1 -- M1 enters both outputs of a pair with weight +1 -- so the epilogue has no loads:
    // per-row bias loads there are dependent round trips that nothing covers (measured: 12 % of the kernel).
    if (tid < BM) bias_s[tid] = (p.bias && m_base + tid < p.Cout_g) ? p.bias[m_base + tid] : 0.f;
    lds_barrier();

    if (tid >= NC) {
        // ================= producers =================
        const int pt = tid - NC;
        // weight offsets of this thread's float4 slots inside a chunk's stage: slot = (tap, parity, m)
        unsigned woff[WR];
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = pt + e * 256;
            const int slab = idx4 / BM, m = idx4 - slab * BM;           // slab = tap * 2 + parity
            const int tap = slab >> 1, par = slab & 1;
            const bool ok = idx4 < WS4 && m_base + m < p.Mpad;
            // packed image: [tap][Cin_pad / 8][2][Mpad] float4
            woff[e] = ok ? 16u * (unsigned)(((tap * (p.Cin_pad >> 3)) * 2 + par) * p.Mpad + m_base + m) : kBufOob;
        }
        // patch items of this thread: (parity, input row, pair j), item ids pt + 256 e
        const float* xg = p.x;
        unsigned coff[NI][4];
        bool wide = false;             // the tile's 66 input columns all exist: an item's four columns are ONE (dword-aligned) 16-byte load
        auto place = [&](int tile) {   // this thread's patch offsets inside tile `tile`
            const int tw_i = tile % p.tiles_w, th_i = (tile / p.tiles_w) % p.tiles_h, n = tile / (p.tiles_w * p.tiles_h);
            const int w0 = tw_i * (2 * kWinoPairs), h0 = th_i * ROWS;
            xg = p.x + (long)n * p.x_sn;
            wide = w0 >= 1 && w0 + 2 * kWinoPairs < p.W && !(kAblate && (p.dbg & 512));
#pragma unroll
            for (int e = 0; e < NI; ++e) {
                const int item = pt + e * 256;
                const int par = item / PLANE, rem = item - par * PLANE;
                const int r = rem / kWinoPairs, j = rem - r * kWinoPairs;
                const int hin = h0 - 1 + r;
                const bool row_ok = item < 2 * PLANE && hin >= 0 && hin < p.H && tile < ntiles;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int win = w0 - 1 + 2 * j + k;
                    coff[e][k] = (row_ok && win >= 0 && win < p.W) ? 4u * (unsigned)(par * p.x_sc + hin * p.x_sh + win) : kBufOob;
                }
            }
        };
        const unsigned ch2 = 8u * (unsigned)p.x_sc;          // byte distance of two channels: element e -> e + 1 of a quad
        // One stage of loads is in flight (issued behind the commit of the previous one, a whole stage of MFMAs to land; a second set
        // two stages ahead measured no different).  Where the loads are conditional -- three forms below -- the compiler's in-order vmcnt
        // bookkeeping falls back to full waits, which is what a commit needs anyway.
        struct Staged { float4 wv[WR]; float d[NI][4][4]; };   // [item][channel e][column k]
        Staged A;
        int lk = 0, lc = 0;                                  // load cursor: tile index of this workgroup, chunk
        auto load = [&](Staged& t) {
            // past the last stage the loads are still issued, against empty buffers (no traffic, zeros)
            const bool live = lk < my_tiles;
            if (live && lc == 0) place(first + slot + lk * slots);
            const long wbase = (long)lc * 2 * p.Mpad * 4;    // floats: chunk c of every tap starts (c * 2 * Mpad) float4 in
            const BufRsrc wb = make_buf(p.w3 + wbase, live ? (unsigned)lmin(((long)12 * p.Cin_pad * p.Mpad - wbase) * 4, 0x7fffffffL) : 0u);
#pragma unroll
            for (int e = 0; e < WR; ++e) t.wv[e] = (kAblate && (p.dbg & 32)) ? make_float4(0.f, 0.f, 0.f, 0.f) : buf_load_f32x4(wb, woff[e]);
            const long left = (long)(p.Cin_g - lc * 8) * p.x_sc * 4;  // bytes up to the end of the channels: absent channels read 0
            const BufRsrc xb = make_buf(xg + (long)lc * 8 * p.x_sc, live ? (unsigned)lmin(left, 0x7fffffffL) : 0u);
            // A whole chunk: the channel step rides in the scalar offset (not range checked: fine, the channels exist) -- a producer
            // VALU instruction takes issue slots from the consumer wave on its SIMD.  Interior tile: one 16-byte load per (item,
            // channel) instead of four dword loads; returning loads, too, slow the consumer waves down (ablation).
            if (lc * 8 + 8 <= p.Cin_g && wide && !(kAblate && (p.dbg & (4 | 128)))) {
#pragma unroll
                for (int it = 0; it < NI; ++it)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 q = buf_load_f32x4_s(xb, coff[it][0], (unsigned)e * ch2);
                        t.d[it][e][0] = q.x; t.d[it][e][1] = q.y; t.d[it][e][2] = q.z; t.d[it][e][3] = q.w;
                    }
            } else if (lc * 8 + 8 <= p.Cin_g) {
#pragma unroll
                for (int it = 0; it < NI; ++it)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (kAblate && (p.dbg & 4)) t.d[it][e][k] = 0.f;
                            else if (kAblate && (p.dbg & 128)) t.d[it][e][k] = __builtin_bit_cast(float, coff[it][k] & 0x3fffffu);   // no memory, the transform stays
                            else t.d[it][e][k] = buf_load_f32_s(xb, coff[it][k], (unsigned)e * ch2);
            } else {
#pragma unroll
                for (int it = 0; it < NI; ++it)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int k = 0; k < 4; ++k) t.d[it][e][k] = buf_load_f32(xb, coff[it][k] + (unsigned)e * ch2);
            }
            if (++lc == nchunk) { lc = 0; ++lk; }
        };
        auto commit = [&](int g, const Staged& t) {   // g: stage counter of this workgroup (buffer g & 1)
            float4* ws = ws0 + (g & 1) * WS4;
            if (kAblate && (p.dbg & 8)) return;
#pragma unroll
            for (int e = 0; e < WR; ++e)
                if (pt + e * 256 < WS4) ws[pt + e * 256] = t.wv[e];
            float4* xs = xs0 + (g & 1) * XS4;
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int item = pt + it * 256;
                if (item >= 2 * PLANE) continue;
                const int par = item / PLANE, rem = item - par * PLANE;
                float v[4][4];                               // [point][channel e]
#ifndef AICG_EMULATED
                if (kAblate && (p.dbg & 256)) {              // the loads are waited for and dropped: no transform
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int k = 0; k < 4; ++k) asm volatile("" :: "v"(t.d[it][e][k]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) xs[(par * 4 + q) * PLANE + rem] = make_float4(0.f, 0.f, 0.f, 0.f);
                    continue;
                }
#endif
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[0][e] = t.d[it][e][0] - t.d[it][e][2];
                    v[1][e] = t.d[it][e][1] + t.d[it][e][2];
                    v[2][e] = t.d[it][e][2] - t.d[it][e][1];
                    v[3][e] = t.d[it][e][1] - t.d[it][e][3];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)                  // plane (parity, point q): [row][pair] = rem
                    xs[(par * 4 + q) * PLANE + rem] = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
            }
        };
        const int total = my_tiles * nchunk;
        if (total == 0) return;
        load(A);
        for (int g = 0; g < total; ++g) {
            commit(g, A);
            load(A);
            lds_barrier();  // stage g published (and the consumers are done with stage g - 1)
        }
        return;
    }

    // ================= consumers =================
    const int lane = tid & 63, wave = tid >> 6;
    int g = 0;
#ifndef AICG_EMULATED
    long long clk0 = 0, wall0 = 0;   // ablation bit 64: shader-clock cycles and 100 MHz ticks of workgroup 0, left in y[0], y[1]
    if (kAblate && (p.dbg & 64)) { clk0 = clock64(); wall0 = wall_clock64(); }
#endif
    if constexpr (!M16) {
        const int half = lane >> 5, l31 = lane & 31;
        f32x16 acc[4][TM];
        for (int k = 0; k < my_tiles; ++k) {
            const int tile = first + slot + k * slots;
            const int tw_i = tile % p.tiles_w, th_i = (tile / p.tiles_w) % p.tiles_h, n = tile / (p.tiles_w * p.tiles_h);
            const int w0 = tw_i * (2 * kWinoPairs), h0 = th_i * ROWS;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {   // accumulator rows r4 * 4 .. + 3 are output channels i * 32 + 8 r4 + 4 half .. + 3
                    const float4 bv = *reinterpret_cast<const float4*>(bias_s + i * 32 + 8 * r4 + 4 * per_tile_lane(half));
                    acc[1][i][r4 * 4 + 0] = bv.x; acc[1][i][r4 * 4 + 1] = bv.y; acc[1][i][r4 * 4 + 2] = bv.z; acc[1][i][r4 * 4 + 3] = bv.w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[0][i][r4 * 4 + r] = acc[2][i][r4 * 4 + r] = acc[3][i][r4 * 4 + r] = 0.f;
                }
            for (int c = 0; c < nchunk; ++c, ++g) {
                lds_barrier();  // stage g is in LDS
                if (kAblate && (p.dbg & 1)) continue;
                // B fragments: plane (half, point q), input row wave + kh, pair l31;  A fragments: slab (tap, half), row i * 32 + l31
                const float4* xq = xs0 + (g & 1) * XS4 + (half * 4) * PLANE + wave * kWinoPairs + l31;
                const float4* wq = ws0 + (g & 1) * WS4 + half * BM + l31;
                // One step = the four MFMAs (k = 4 channels of this lane half) of tap t = kh * 4 + q on row block i.  Steps run in units of U
                // with different accumulators, their MFMAs interleaved, and a unit's fragments are read while the previous unit computes,
                // in an order the scheduler may not change: left to itself it gathers the reads into bursts and the matrix pipe drains
                // while a burst lands.  (What remains is the hand-over cost DESIGN 2.1 measured: ~74 cycles per MFMA instead of 64.)
                constexpr int S = 12 * TM, U = TM >= 2 ? TM : 2;
                float4 af[S], bf[12];
                auto fetch = [&](int st) {
                    const int t = st / TM, i = st - t * TM;
                    const bool one = kAblate && (p.dbg & 2);
                    af[st] = wq[one ? 0 : t * 2 * BM + i * 32];
                    if (i == 0) bf[t] = xq[one ? 0 : (t & 3) * PLANE + (t >> 2) * kWinoPairs];
                };
#pragma unroll
                for (int st = 0; st < U; ++st) fetch(st);
#pragma unroll
                for (int u0 = 0; u0 < S; u0 += U) {
#pragma unroll
                    for (int st = u0 + U; st < u0 + 2 * U; ++st)
                        if (st < S) fetch(st);
                    sched_fence();
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int st = u0; st < u0 + U; ++st) {
                            const int t = st / TM, i = st - t * TM, q = t & 3;
                            const float av = e == 0 ? af[st].x : e == 1 ? af[st].y : e == 2 ? af[st].z : af[st].w;
                            const float bv = e == 0 ? bf[t].x : e == 1 ? bf[t].y : e == 2 ? bf[t].z : bf[t].w;
                            acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[q][i], 0, 0, 0);
                        }
                    sched_fence();
                }
            }
            // ---- epilogue: Y(2 j) = M0 + M1 + M2, Y(2 j + 1) = M1 - M2 - M3 (bias inside M1), activation; lane j stores the pair as one float2
            const int ho = h0 + wave, wo = w0 + 2 * l31;
            if (tile >= ntiles || ho >= p.Ho || wo >= p.Wo || (kAblate && (p.dbg & 16))) continue;
            const bool pair_ok = wo + 1 < p.Wo;
            float* yrow = p.y + (long)n * p.y_sn + (long)ho * p.y_sh + wo;
            const long y_sc = per_tile(p.y_sc);
            const int row0 = per_tile_lane(m_base + 4 * half);
            const bool al8 = ((p.y_sn | p.y_sc | p.y_sh) & 1) == 0 && ((uintptr_t)p.y & 7) == 0;
            auto body = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (m >= p.Cout_g) continue;
                        float y0 = (acc[0][i][r] + acc[1][i][r]) + acc[2][i][r];
                        float y1 = (acc[1][i][r] - acc[2][i][r]) - acc[3][i][r];
                        y0 = act_static<ACT>(y0, p.act, p.act_slope);
                        y1 = act_static<ACT>(y1, p.act, p.act_slope);
                        float* dst = yrow + (long)m * y_sc;
                        if (pair_ok && al8) *reinterpret_cast<float2*>(dst) = make_float2(y0, y1);
                        else { dst[0] = y0; if (pair_ok) dst[1] = y1; }
                    }
            };
            if (p.act == AICG_ACT_NONE) body(std::integral_constant<int, 0>{});
            else if (p.act == AICG_ACT_RELU) body(std::integral_constant<int, 1>{});
            else body(std::integral_constant<int, 3>{});
        }
    } else {
        // ---- 48 rows on 16 x 16 x 4: lane = (k slot ks, index l15); row blocks rb = 0..2 (channel rb * 16 + l15 of A, channels
        // rb * 16 + 4 ks + r of the accumulator), pair blocks cb = 0..1 (pair cb * 16 + l15)
        const int ks = lane >> 4, l15 = lane & 15;
        const int par = ks & 1, e2 = ks >> 1;            // this slot's channels of a chunk: 2 (2 e2 + s) + par for k-step s = 0, 1
        f32x4v acc[4][3][2];
        for (int k = 0; k < my_tiles; ++k) {
            const int tile = first + slot + k * slots;
            const int tw_i = tile % p.tiles_w, th_i = (tile / p.tiles_w) % p.tiles_h, n = tile / (p.tiles_w * p.tiles_h);
            const int w0 = tw_i * (2 * kWinoPairs), h0 = th_i * ROWS;
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const float4 bv = *reinterpret_cast<const float4*>(bias_s + rb * 16 + 4 * per_tile_lane(ks));
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    acc[1][rb][cb][0] = bv.x; acc[1][rb][cb][1] = bv.y; acc[1][rb][cb][2] = bv.z; acc[1][rb][cb][3] = bv.w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[0][rb][cb][r] = acc[2][rb][cb][r] = acc[3][rb][cb][r] = 0.f;
                }
            }
            for (int c = 0; c < nchunk; ++c, ++g) {
                lds_barrier();  // stage g is in LDS
                if (kAblate && (p.dbg & 1)) continue;
                // 8-byte fragments out of the quads: float2 index = 2 x float4 index + e2
                const float2* xq = reinterpret_cast<const float2*>(xs0 + (g & 1) * XS4 + (par * 4) * PLANE + wave * kWinoPairs + l15) + e2;
                const float2* wq = reinterpret_cast<const float2*>(ws0 + (g & 1) * WS4 + par * BM + l15) + e2;
                float2 af[12][3], bf[12][2];
                auto fetch = [&](int t) {
                    const bool one = kAblate && (p.dbg & 2);
#pragma unroll
                    for (int rb = 0; rb < 3; ++rb) af[t][rb] = wq[one ? 0 : 2 * (t * 2 * BM + rb * 16)];
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) bf[t][cb] = xq[one ? 0 : 2 * ((t & 3) * PLANE + (t >> 2) * kWinoPairs + cb * 16)];
                };
                fetch(0);
#pragma unroll
                for (int t = 0; t < 12; ++t) {   // tap t = kh * 4 + q: 12 MFMAs on six accumulators, the next tap's five reads in flight
                    if (t + 1 < 12) fetch(t + 1);
                    sched_fence();
                    const int q = t & 3;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int rb = 0; rb < 3; ++rb)
                                acc[q][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(s ? af[t][rb].y : af[t][rb].x, s ? bf[t][cb].y : bf[t][cb].x,
                                                                                        acc[q][rb][cb], 0, 0, 0);
                    sched_fence();
                }
            }
            const int ho = h0 + wave;
            if (tile >= ntiles || ho >= p.Ho || (kAblate && (p.dbg & 16))) continue;
            float* yrow = p.y + (long)n * p.y_sn + (long)ho * p.y_sh;
            const long y_sc = per_tile(p.y_sc);
            const int row0 = per_tile_lane(m_base + 4 * ks);
            const bool al8 = ((p.y_sn | p.y_sc | p.y_sh) & 1) == 0 && ((uintptr_t)p.y & 7) == 0;
            auto body = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int wo = w0 + 2 * (cb * 16 + l15);
                    if (wo >= p.Wo) continue;
                    const bool pair_ok = wo + 1 < p.Wo;
#pragma unroll
                    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = row0 + rb * 16 + r;
                            if (m >= p.Cout_g) continue;
                            float y0 = (acc[0][rb][cb][r] + acc[1][rb][cb][r]) + acc[2][rb][cb][r];
                            float y1 = (acc[1][rb][cb][r] - acc[2][rb][cb][r]) - acc[3][rb][cb][r];
                            y0 = act_static<ACT>(y0, p.act, p.act_slope);
                            y1 = act_static<ACT>(y1, p.act, p.act_slope);
                            float* dst = yrow + (long)m * y_sc + wo;
                            if (pair_ok && al8) *reinterpret_cast<float2*>(dst) = make_float2(y0, y1);
                            else { dst[0] = y0; if (pair_ok) dst[1] = y1; }
                        }
                }
            };
            if (p.act == AICG_ACT_NONE) body(std::integral_constant<int, 0>{});
            else if (p.act == AICG_ACT_RELU) body(std::integral_constant<int, 1>{});
            else body(std::integral_constant<int, 3>{});
        }
    }
#ifndef AICG_EMULATED
    if (kAblate && (p.dbg & 64) && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        p.y[0] = (float)(clock64() - clk0);
        p.y[1] = (float)(wall_clock64() - wall0);
    }
#endif
}

// returns 0 launched, < 0 error, 1 not applicable.  p.w3 must point at the k8-interleaved image of the (Cout, Cin, 3, 4) Winograd
// kernel (taps = 12), p.Mpad / p.Cin_pad describe that image.
template <int BM, int ROWS>
static int launch_conv_ws3w(ConvArgs& p, hipStream_t stream) {
    p.tiles_w = idiv_up(p.Wo, 2 * kWinoPairs);
    p.tiles_h = idiv_up(p.Ho, ROWS);
    p.nchunk = idiv_up(p.Cin_g, 8);
    const size_t lds = (size_t)(2 * 2 * 4 * (ROWS + 2) * kWinoPairs + 2 * 24 * BM) * sizeof(float4) + BM * sizeof(float);
    const bool off_ok = (long)16 * p.x_sc + (long)p.H * p.x_sh < (1L << 29) && (long)12 * p.Cin_pad * p.Mpad < (1L << 29);
    if (lds > 160 * 1024 || !off_ok) return 1;
    const long ntiles = (long)p.N * p.tiles_h * p.tiles_w;
    if (ntiles > 2147483647L - 8) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    // one workgroup per CU: 8 XCDs x `slots` workgroups per output-channel tile, each walking its share of the tile list
    const int my = idiv_up(p.Cout_g, BM);
    const int per_xcd = (int)((ntiles + 7) >> 3);
    int slots = 32 / my;
    if (slots < 1) slots = 1;
    if (slots > per_xcd) slots = per_xcd;
    dim3 grid((unsigned)(8 * slots), (unsigned)my, 1);
    allow_dynamic_lds((const void*)conv_ws3w_kernel<BM, ROWS>, lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_ws3w_kernel<BM, ROWS>), grid, dim3(64 * ROWS + 256), lds, stream, p);
    return check_launch("conv_ws3w_kernel");
}

int run_ws3w_96(ConvArgs& p, hipStream_t st);
int run_ws3w_64(ConvArgs& p, hipStream_t st);
int run_ws3w_48(ConvArgs& p, hipStream_t st);
int run_ws3w_32(ConvArgs& p, hipStream_t st);

}  // namespace aicg
