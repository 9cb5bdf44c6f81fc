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
// K = Cin_g*KH*KW walked as [channel chunk][tap][channel in chunk].
// A (weights) is pre-packed on the host as [tap][Cin_pad][Mpad] so a stage is a few coalesced float4 row copies into LDS;
// B is never materialised: a chunk of BKC input channels of the input patch (with halo) is staged
// into LDS once -- with the fused pre-activation applied once per element, not once per tap -- and
// every tap reads it at a shifted offset.  Lane l of a wave feeds the MFMA with
// A[k = l>>5][m = l&31] and B[k = l>>5][n = l&31]: both are unit-stride, conflict-free ds_read_b32.
//
// This header holds the kernel templates and their launchers; the instantiations are spread over conv_ws_*.hip /
// conv_ws16.hip / conv_single_role.hip (one hipcc job each: the family takes minutes to compile as one unit) and
// conv.hip keeps the dispatcher behind aicg_conv_forward.
#pragma once
#include <type_traits>

#include "common.h"

#include <cstdint>
#include <cstdlib>

namespace aicg {

// Phase-ablation switches (AICG_CONV_ABLATE bits in ConvArgs::dbg) exist only in builds made with -DAICG_CONV_ABLATION (build.py --dev)
// (tools/): the shipped kernels carry no profiling branches in their hot loops.
#ifdef AICG_CONV_ABLATION
static constexpr bool kAblate = true;
#else
static constexpr bool kAblate = false;
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x;
    const float* w;
    const float* w3;   // k8-interleaved image of the same weights (conv_ws3.h), or nullptr
    const float* wsplit;   // bf16 hi / lo image (conv_ws3s.h) when the layer runs in split precision, or nullptr
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
    int shuffle;   // 2: GEMM row m / position (ho, wo) -> y[m >> 2][2 ho + ((m >> 1) & 1)][2 wo + (m & 1)] (k = s = 2 ConvTranspose2d)
    int res_mul;   // the residual operand multiplies (U-Net multiplicative skip) instead of adding, after the activation
    // derived tiling
    int TW, TWlog2, TH, TH_in, TW_in, TWp, CHS, BKC, BKClog2, TT, tiles_w, tiles_h, nchunk, taps, Mpad, Cin_pad, xs_elems, xs_total;
    unsigned div_chs, div_twp;  // ceil(2^32 / d) multipliers: idx / d == umulhi(idx, mul) for idx * d < 2^32
    long w_group_stride;
    // Phase stagger (wave-specialised kernels, two workgroups per CU): the consumers of the workgroup that shares a CU with an
    // earlier one start `stagger` shader cycles late, once, in the launch's initial fill (flat block id < stagger_first).
    // Identical tiles that start together stay in lockstep for the whole launch -- both in their prologue, both in their K
    // loop halving the matrix pipe, both in their epilogue with the pipe idle (r2 in-kernel timeline: 17 % of a 96-row MDX
    // tile's lifetime); half a tile out of phase, one workgroup's prologue / epilogue runs under the other's MFMAs.
    int stagger, stagger_first;
    int wide_ok;   // y (and res) rows are 16-byte aligned and TW % 4 == 0: interior tiles may use the float4 epilogue
    int f16;  // aicg_conv_desc.split == 2: fp16 operands on the matrix pipe where the layer's kernel has that form (conv_g1.h, conv_g1w.h)
    int dbg;  // AICG_CONV_ABLATE bits (profiling only): 1 no global loads, 2 no LDS commit, 4 no barriers, 8 no MFMA loop, 16 no epilogue
};

// output / residual element of GEMM row cg (= g * Cout_g + m) at position (ho, wo), relative to the image base
__device__ __forceinline__ long out_index(const ConvArgs& p, int cg, int ho, int wo, long sc, long sh) {
    return p.shuffle ? (long)(cg >> 2) * sc + (long)(2 * ho + ((cg >> 1) & 1)) * sh + 2 * wo + (cg & 1)
                     : (long)cg * sc + (long)ho * sh + wo;
}
// y = [y_old +] out_scale * (act(v [+ r]) [+ r | * r])
__device__ __forceinline__ float combine(const ConvArgs& p, float v, float r, float y_old) {
    if (p.res_first) v += r;
    v = apply_act(v, p.act, p.act_slope);
    if (!p.res_first) v = p.res_mul ? v * r : v + r;
    return v * p.out_scale + y_old;
}

// Optional in-kernel timeline (tools/conv_trace.py builds a private copy of the library with -DAICG_CONV_TRACE): wave 0 of the
// consumers of the first kTraceWgs workgroups records s_memtime at entry, first stage ready, end of the K loop, end of the
// epilogue, plus the hardware id; every `kTraceEvery`-th stage start as well.  Never compiled into the shipped library.
#ifdef AICG_CONV_TRACE
static constexpr int kTraceWgs = 8192, kTraceSlots = 8;
extern __device__ unsigned long long g_conv_trace[kTraceWgs * kTraceSlots];
__device__ __forceinline__ void trace_mark(int wg, int slot) {
    if (wg < kTraceWgs && (threadIdx.x & 63) == 0) g_conv_trace[wg * kTraceSlots + slot] = __builtin_readcyclecounter();
}
__device__ __forceinline__ void trace_val(int wg, int slot, unsigned long long v) {
    if (wg < kTraceWgs && (threadIdx.x & 63) == 0) g_conv_trace[wg * kTraceSlots + slot] = v;
}
// per-stage detail for workgroups 0 .. kTrace2Wgs-1: [wg][role 0 consumer / 1 producer][stage < 32][event]
static constexpr int kTrace2Wgs = 64;
extern __device__ unsigned long long g_conv_trace2[kTrace2Wgs * 2 * 32 * 4];
__device__ __forceinline__ void trace2(int wg, int role, int stage, int ev) {
    if (wg < kTrace2Wgs && stage < 32 && (threadIdx.x & 63) == 0)
        g_conv_trace2[((wg * 2 + role) * 32 + stage) * 4 + ev] = __builtin_readcyclecounter();
}
#else
__device__ __forceinline__ void trace_mark(int, int) {}
__device__ __forceinline__ void trace_val(int, int, unsigned long long) {}
__device__ __forceinline__ void trace2(int, int, int, int) {}
#endif

// Epilogue activation with the common cases resolved at compile time (ACT 0 none, 1 ReLU, 2 leaky ReLU, 3 any: run-time code,
// 4 GELU inline -- conv_g1.h only: ~25 instructions per value where the out-of-line call drains the memory counters per value)
template <int ACT>
__device__ __forceinline__ float act_static(float v, int act, float slope) {
    if (ACT == 0) return v;
    if (ACT == 1) return v > 0.f ? v : 0.f;
    if (ACT == 2) return v > 0.f ? v : v * slope;
    if (ACT == 4) return gelu_erf(v);
    return apply_act(v, act, slope);
}
// epi(full_tag, act_tag): one call, chosen by the layer's activation and by whether the tile lies wholly inside the output
template <typename F>
__device__ __forceinline__ void dispatch_epilogue(int act, bool interior, F&& epi) {
    using T = std::true_type;
    using N = std::false_type;
    if (interior) {
        if (act == AICG_ACT_NONE) epi(T{}, std::integral_constant<int, 0>{});
        else if (act == AICG_ACT_RELU) epi(T{}, std::integral_constant<int, 1>{});
        else if (act == AICG_ACT_LRELU) epi(T{}, std::integral_constant<int, 2>{});
        else epi(T{}, std::integral_constant<int, 3>{});
    } else {
        if (act == AICG_ACT_NONE) epi(N{}, std::integral_constant<int, 0>{});
        else if (act == AICG_ACT_RELU) epi(N{}, std::integral_constant<int, 1>{});
        else if (act == AICG_ACT_LRELU) epi(N{}, std::integral_constant<int, 2>{});
        else epi(N{}, std::integral_constant<int, 3>{});
    }
}

static constexpr int KSTAGE = 64;  // max K rows of packed weights staged per barrier pair

// Software pipeline (guide T14, "issue early / write late"):
//   global loads of stage s+1 (weights, and the input patch when s+1 opens a new channel chunk) are issued into
//   registers right after the barrier that publishes stage s in LDS, fly under the MFMA loop of stage s, and are
//   written to LDS after the next barrier.  Inside the MFMA loop the A/B fragments of step i+1 are read from LDS
//   before the MFMAs of step i issue.
template <int BM, int BN, int WM, int WN, int XR>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN == 4 ? 2 : 4)) conv_mfma_kernel(ConvArgs p) {
    constexpr int NT = 64 * WM * WN;                   // 4 or 8 waves per workgroup
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    constexpr int WR = (KSTAGE * BM / 4 + NT - 1) / NT;  // float4 weight loads per thread per stage
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
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
    const int stages_per_chunk = (p.taps + p.TT - 1) / p.TT;
    const int nstages = p.nchunk * stages_per_chunk;

    float xv[XR];
    float4 wv[WR];

    // ---- issue the global loads of one stage (no waits here) --------------------------------------------------
    auto prefetch = [&](int st) {
        const int c = st / stages_per_chunk;
        const int tap0 = (st - c * stages_per_chunk) * p.TT;
        const int rows = imin(p.TT, p.taps - tap0) << p.BKClog2;
        // packed weights: [tap][Cin_pad][Mpad]; stage row r = (tap tt = r / BKC, channel c*BKC + r % BKC)
        const float* wrow0 = wg + ((long)tap0 * p.Cin_pad + (long)c * p.BKC) * p.Mpad;
        if (kAblate && (p.dbg & 1)) return;
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = tid + e * NT;
            const int r = idx4 / (BM / 4);
            const int c4 = idx4 - r * (BM / 4);
            const int mcol = m_base + c4 * 4;
            const int tt = r >> p.BKClog2, ci = r & (p.BKC - 1);
            // branch-free: out-of-range slots re-read the first row and are zeroed afterwards
            const bool ok = r < rows && mcol < p.Mpad;
            const long off = ok ? ((long)tt * p.Cin_pad + ci) * p.Mpad + mcol : 0;
            const float4 t = *reinterpret_cast<const float4*>(wrow0 + off);
            wv[e] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tap0 == 0) {  // a new channel chunk: its input patch (halo included)
#pragma unroll
            for (int e = 0; e < XR; ++e) {
                const int idx = tid + e * NT;
                const int ci = (int)__umulhi((unsigned)idx, p.div_chs);
                const int rem = idx - ci * p.CHS;
                const int r = (int)__umulhi((unsigned)rem, p.div_twp);
                const int col = rem - r * p.TWp;
                const int cg = c * p.BKC + ci;
                const int hin = hin0 + r, win = win0 + col;
                const bool ok = idx < p.xs_total && cg < p.Cin_g && col < p.TW_in && hin >= 0 && hin < p.H && win >= 0 && win < p.W;
                const long off = ok ? (long)cg * p.x_sc + (long)hin * p.x_sh + win : 0;  // branch-free zero padding
                const float t = xg[off];
                xv[e] = ok ? t : 0.f;
            }
        }
    };
    // ---- registers -> LDS (pre-activation fused once per staged element) ---------------------------------------
    auto commit = [&](int st) {
        const int c = st / stages_per_chunk;
        const int tap0 = (st - c * stages_per_chunk) * p.TT;
        if (kAblate && (p.dbg & 2)) return;
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = tid + e * NT;
            if (idx4 < KSTAGE * (BM / 4)) *reinterpret_cast<float4*>(ws + idx4 * 4) = wv[e];
        }
        if (tap0 == 0) {
#pragma unroll
            for (int e = 0; e < XR; ++e) {
                const int idx = tid + e * NT;
                if (idx < p.xs_total) xs[idx] = apply_act(xv[e], p.pre_act, p.pre_slope);
            }
        }
    };

    prefetch(0);
    for (int st = 0; st < nstages; ++st) {
        if (!(kAblate && (p.dbg & 4))) __syncthreads();  // every wave has finished the MFMAs of the previous stage: LDS may be overwritten
        commit(st);
        if (!(kAblate && (p.dbg & 4))) __syncthreads();
        if (st + 1 < nstages) prefetch(st + 1);  // in flight during the MFMA loop below
        if (kAblate && (p.dbg & 8)) continue;

        const int c = st / stages_per_chunk;
        const int tap0 = (st - c * stages_per_chunk) * p.TT;
        const int nt = imin(p.TT, p.taps - tap0);
        int kh0 = tap0 / p.KW, kw0 = tap0 - kh0 * p.KW;  // first tap of this stage
        const int nsteps = nt * (p.BKC >> 1);  // MFMA k-steps of this stage
        const float* wt = ws + a_off;
        // Two fragment register sets used alternately (even / odd k-step): the LDS reads of step s+1 are issued
        // before the MFMAs of step s, and each MFMA group waits only for its own (older) reads -- the in-order
        // lgkm counter lets the newer reads stay in flight.
        float a0[TM], b0[TN], a1[TM], b1[TN];
        int kh = kh0, kw = kw0, kk = 0;  // (tap, channel pair) of the step being fetched
        auto fetch = [&](float (&a)[TM], float (&b)[TN], int s) {
            const float* xt = xs + kh * p.dh * p.TWp + kw * p.dw + kk * p.CHS;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = wt[s * 2 * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = xt[boff[j]];
            kk += 2;
            if (kk == p.BKC) { kk = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
        };
        auto mma = [&](float (&a)[TM], float (&b)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        };
        fetch(a0, b0, 0);
        int s = 0;
        for (; s + 2 <= nsteps; s += 2) {
            fetch(a1, b1, s + 1);
            mma(a0, b0);
            fetch(a0, b0, s + 2);  // unconditional: past the last step this reads (never uses) the LDS slack rows
            mma(a1, b1);
        }
        if (s < nsteps) mma(a0, b0);  // odd step count (BKC = 2 with an odd number of taps)
    }

    // ---- epilogue: y = [y +] out_scale * (act(acc + bias [+ res]) [+ res]) ------------------------------------
    // res / y may alias (in-place residual), so the compiler cannot move a load across a store: all operand loads
    // of a 32x32 tile are issued first, then the 16 results per lane are formed and stored.
    if (kAblate && (p.dbg & 16)) { if (acc[0][0][0] != 12345.f) return; }
    const long y_base = (long)n * p.y_sn, r_base = (long)n * p.r_sn;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (TN * 32) + j * 32 + l31;
        const int ho = h0 + (nl >> p.TWlog2), wo = w0 + (nl & (p.TW - 1));
        const bool col_ok = ho < p.Ho && wo < p.Wo;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m0 = m_base + wm * (TM * 32) + i * 32 + 4 * half;
            float rv[16], yv[16], bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2);
                const bool ok = col_ok && m < p.Cout_g;
                const int co = g * p.Cout_g + m;
                bv[r] = (ok && p.bias) ? p.bias[co] : 0.f;
                rv[r] = (ok && p.res) ? p.res[r_base + out_index(p, co, ho, wo, p.r_sc, p.r_sh)] : 0.f;
                yv[r] = (ok && p.accumulate) ? p.y[y_base + out_index(p, co, ho, wo, p.y_sc, p.y_sh)] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2);
                if (!(col_ok && m < p.Cout_g)) continue;
                const int co = g * p.Cout_g + m;
                p.y[y_base + out_index(p, co, ho, wo, p.y_sc, p.y_sh)] = combine(p, acc[i][j][r] + bv[r], rv[r], yv[r]);
            }
        }
    }
}

// ---- narrow-M variant on v_mfma_f32_16x16x4_f32 ------------------------------------------------------------------------
// Layers with 16 or 48 output channels (RMVPE level 0, the 48-channel first level of the MDX-Net U-Net) waste 50 % / 25 % of a
// 32-row MFMA tile; the 16x16x4 instruction has the same FLOP rate (32-cycle issue) and tiles M in steps of 16.
// Same staging pipeline as conv_mfma_kernel; A[m = l&15][k = l>>4], B[k = l>>4][n = l&15], D: col = l&15, row = 4*(l>>4) + reg.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BM, int XR>
__global__ void __launch_bounds__(256, 2) conv_mfma16_kernel(ConvArgs p) {
    constexpr int NT = 256, BN = 256;
    constexpr int TM = BM / 16;   // 16-row tiles per wave (every wave covers all BM rows)
    constexpr int TN = 4;         // 16-column tiles per wave: 64 positions per wave, 4 waves
    constexpr int WR = (KSTAGE * BM / 4 + NT - 1) / NT;
    HIP_DYNAMIC_SHARED(float, smem)
    float* xs = smem;
    float* ws = smem + p.xs_elems;
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
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
        const int nl = wn * 64 + j * 16 + r16;
        const int jh = nl >> p.TWlog2, jw = nl & (p.TW - 1);
        boff[j] = jh * p.sh * p.TWp + jw * p.sw + q * p.CHS;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const float* xg = p.x + (long)n * p.x_sn + (long)g * p.Cin_g * p.x_sc;
    const float* wg = p.w + (long)g * p.w_group_stride;
    const int hin0 = h0 * p.sh - p.ph, win0 = w0 * p.sw - p.pw;
    const int stages_per_chunk = (p.taps + p.TT - 1) / p.TT;
    const int nstages = p.nchunk * stages_per_chunk;
    float xv[XR];
    float4 wv[WR];
    auto prefetch = [&](int st) {
        const int c = st / stages_per_chunk;
        const int tap0 = (st - c * stages_per_chunk) * p.TT;
        const int rows = imin(p.TT, p.taps - tap0) << p.BKClog2;
        const float* wrow0 = wg + ((long)tap0 * p.Cin_pad + (long)c * p.BKC) * p.Mpad;
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = tid + e * NT;
            const int r = idx4 / (BM / 4);
            const int c4 = idx4 - r * (BM / 4);
            const int mcol = m_base + c4 * 4;
            const int tt = r >> p.BKClog2, ci = r & (p.BKC - 1);
            const bool ok = r < rows && mcol < p.Mpad;
            const long off = ok ? ((long)tt * p.Cin_pad + ci) * p.Mpad + mcol : 0;
            const float4 t = *reinterpret_cast<const float4*>(wrow0 + off);
            wv[e] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tap0 == 0) {
#pragma unroll
            for (int e = 0; e < XR; ++e) {
                const int idx = tid + e * NT;
                const int ci = (int)__umulhi((unsigned)idx, p.div_chs);
                const int rem = idx - ci * p.CHS;
                const int r = (int)__umulhi((unsigned)rem, p.div_twp);
                const int col = rem - r * p.TWp;
                const int cg = c * p.BKC + ci;
                const int hin = hin0 + r, win = win0 + col;
                const bool ok = idx < p.xs_total && cg < p.Cin_g && col < p.TW_in && hin >= 0 && hin < p.H && win >= 0 && win < p.W;
                const long off = ok ? (long)cg * p.x_sc + (long)hin * p.x_sh + win : 0;
                const float t = xg[off];
                xv[e] = ok ? t : 0.f;
            }
        }
    };
    auto commit = [&](int st) {
        const int c = st / stages_per_chunk;
        const int tap0 = (st - c * stages_per_chunk) * p.TT;
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = tid + e * NT;
            if (idx4 < KSTAGE * (BM / 4)) *reinterpret_cast<float4*>(ws + idx4 * 4) = wv[e];
        }
        if (tap0 == 0) {
#pragma unroll
            for (int e = 0; e < XR; ++e) {
                const int idx = tid + e * NT;
                if (idx < p.xs_total) xs[idx] = apply_act(xv[e], p.pre_act, p.pre_slope);
            }
        }
    };
    prefetch(0);
    for (int st = 0; st < nstages; ++st) {
        __syncthreads();
        commit(st);
        __syncthreads();
        if (st + 1 < nstages) prefetch(st + 1);
        const int c = st / stages_per_chunk;
        const int tap0 = (st - c * stages_per_chunk) * p.TT;
        const int nt = imin(p.TT, p.taps - tap0);
        int kh = tap0 / p.KW, kw = tap0 - kh * p.KW, kk = 0;
        const int nsteps = nt * (p.BKC >> 2);  // k-steps of 4 rows
        const float* wt = ws + q * BM + r16;
        float a0[TM], b0[TN], a1[TM], b1[TN];
        auto fetch = [&](float (&a)[TM], float (&b)[TN], int s) {
            const float* xt = xs + kh * p.dh * p.TWp + kw * p.dw + kk * p.CHS;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = wt[s * 4 * BM + i * 16];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = xt[boff[j]];
            kk += 4;
            if (kk == p.BKC) { kk = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
        };
        auto mma = [&](float (&a)[TM], float (&b)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        };
        fetch(a0, b0, 0);
        int s = 0;
        for (; s + 2 <= nsteps; s += 2) {
            fetch(a1, b1, s + 1);
            mma(a0, b0);
            fetch(a0, b0, s + 2);
            mma(a1, b1);
        }
        if (s < nsteps) mma(a0, b0);
    }
    const long y_base = (long)n * p.y_sn, r_base = (long)n * p.r_sn;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * 64 + j * 16 + r16;
        const int ho = h0 + (nl >> p.TWlog2), wo = w0 + (nl & (p.TW - 1));
        const bool col_ok = ho < p.Ho && wo < p.Wo;
        float rv[TM * 4], yv[TM * 4], bv[TM * 4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + i * 16 + q * 4 + r;
                const bool ok = col_ok && m < p.Cout_g;
                const int co = g * p.Cout_g + m;
                bv[i * 4 + r] = (ok && p.bias) ? p.bias[co] : 0.f;
                rv[i * 4 + r] = (ok && p.res) ? p.res[r_base + out_index(p, co, ho, wo, p.r_sc, p.r_sh)] : 0.f;
                yv[i * 4 + r] = (ok && p.accumulate) ? p.y[y_base + out_index(p, co, ho, wo, p.y_sc, p.y_sh)] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + i * 16 + q * 4 + r;
                if (!(col_ok && m < p.Cout_g)) continue;
                const int co = g * p.Cout_g + m;
                p.y[y_base + out_index(p, co, ho, wo, p.y_sc, p.y_sh)] = combine(p, acc[i][j][r] + bv[i * 4 + r], rv[i * 4 + r], yv[i * 4 + r]);
            }
    }
}

// ---- wave-specialised variant ------------------------------------------------------------------------------------------
// Ablation of the kernel above on MI355X (MDX-Net level 1, 96 -> 96 channels 3x3, 5.97 ms): MFMA loop alone 4.25 ms, staging +
// epilogue alone 2.27 ms -- co-resident workgroups run in lockstep, so their staging phases coincide and the matrix pipe idles.
// Here 4 producer waves (one per SIMD) stage stage s+1 (global -> registers -> the other LDS buffer) while the WM x WN consumer
// waves run the MFMAs of stage s; one LDS-only workgroup barrier per stage hands the buffers over.  A producer's VALU / VMEM /
// LDS-write instructions co-issue with the consumer wave's MFMAs on the same SIMD.
//   LDS: [patch 0][patch 1][weights 0][weights 1]; the patch buffer alternates per channel chunk, the weight buffer per stage.
//   Producer addressing is hoisted: element e of a thread always maps to the same (channel-in-chunk, row, column) of the patch and
//   the same (tap-in-stage, channel, m) of the weight stage, so its offsets / spatial validity are computed once per workgroup and a
//   stage costs one add + one compare per load.  The regions are padded to whole producer passes: LDS stores are unconditional.
template <int BM, int KS>
struct WsGeom {
    static constexpr int PNT = 256;
    static constexpr int WR = (KS * BM / 4 + PNT - 1) / PNT;
    static constexpr int WS_ELEMS = ((KS + 4) * BM > WR * PNT * 4) ? (KS + 4) * BM : WR * PNT * 4;  // + slack rows: discarded last fragment prefetch
};

// The producer role of the wave-specialised kernels: 256 threads (ptid) stage every K stage of one output tile.
template <int BM, int XR, int KS, bool BOOST>
__device__ __forceinline__ void ws_produce(const ConvArgs& p, float* xs0, float* ws0, int ptid, int n, int g, int h0, int w0,
                                           int m_base, int nstages) {
    constexpr int PNT = 256;
    constexpr int WR = WsGeom<BM, KS>::WR;
    constexpr int WS_ELEMS = WsGeom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = XR * PNT;
    const float* xg = p.x + (long)n * p.x_sn + (long)g * p.Cin_g * p.x_sc;
    const float* wg = p.w + (long)g * p.w_group_stride + m_base;
    if (kAblate && (p.dbg & 1)) {
        for (int st = 0; st < nstages; ++st) lds_barrier();
        return;
    }
    // BOOST: one workgroup per CU (8-consumer shape): nothing else runs while its consumers wait for a late stage
    if (BOOST) __builtin_amdgcn_s_setprio(2);
    // Hoisted addressing: element e of a thread always maps to the same (channel-in-chunk, row, column) of the patch and the
    // same (tap-in-stage, channel, m) of the weight stage, so its BYTE offset relative to the chunk / stage base is computed
    // once per workgroup.  Slots outside the image (zero padding), past the patch or past the weight tile carry kBufOob and
    // read as 0 through the buffer range check; channels the layer does not have fall past num_records of the chunk's buffer.
    unsigned poff[XR], woff[WR];
    {
        const int hin0 = h0 * p.sh - p.ph, win0 = w0 * p.sw - p.pw;
#pragma unroll
        for (int e = 0; e < XR; ++e) {
            const int idx = ptid + e * PNT;
            const int ci = (int)__umulhi((unsigned)idx, p.div_chs);
            const int rem = idx - ci * p.CHS;
            const int r = (int)__umulhi((unsigned)rem, p.div_twp);
            const int col = rem - r * p.TWp;
            const int hin = hin0 + r, win = win0 + col;
            const bool ok = idx < p.xs_total && col < p.TW_in && hin >= 0 && hin < p.H && win >= 0 && win < p.W;
            poff[e] = ok ? 4u * (unsigned)(ci * p.x_sc + hin * p.x_sh + win) : kBufOob;
        }
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = ptid + e * PNT;
            const int r = idx4 / (BM / 4);
            const int c4 = idx4 - r * (BM / 4);
            const int tt = r >> p.BKClog2, ci = r & (p.BKC - 1);
            const bool ok = m_base + c4 * 4 < p.Mpad && r < KS;
            woff[e] = ok ? 4u * (unsigned)((tt * p.Cin_pad + ci) * p.Mpad + c4 * 4) : kBufOob;
        }
    }
    // lds_barrier() does not wait for loads in flight: a stage's load latency spans a whole consumer stage.
    // load() only issues loads -- nothing consumes a loaded value here, so the whole batch goes out back to back.  (The r1
    // kernel selected the padding zeros right behind each 64-bit-addressed global_load; under the 128-register cap hipcc then
    // waited vmcnt(0) after every single patch load: 12 serialised HBM round trips per channel chunk.)
    auto load = [&](int c, int tap0, float4 (&wv)[WR], float (&xv)[XR]) {
        // rows past this stage's K extent (fewer taps in a chunk's last stage) are never multiplied: whatever they read is unused
        // ... but must not run past the group's packed weights: num_records ends the buffer there
        const long wbase = ((long)tap0 * p.Cin_pad + (long)c * p.BKC) * p.Mpad;
        const BufRsrc wb = make_buf(wg + wbase, (unsigned)lmin(((long)p.taps * p.Cin_pad * p.Mpad - wbase - m_base) * 4, 0x7fffffffL));
#pragma unroll
        for (int e = 0; e < WR; ++e) wv[e] = (kAblate && (p.dbg & 512)) ? make_float4(0.f, 0.f, 0.f, 0.f) : buf_load_f32x4(wb, woff[e]);
        if (tap0 == 0) {  // a new channel chunk: its input patch (halo included)
            const long left = (long)(p.Cin_g - c * p.BKC) * p.x_sc * 4;  // bytes up to the end of this group's channels
            const BufRsrc xb = make_buf(xg + (long)c * p.BKC * p.x_sc, (unsigned)lmin(left, 0x7fffffffL));
#pragma unroll
            for (int e = 0; e < XR; ++e) xv[e] = (kAblate && (p.dbg & 256)) ? 0.f : buf_load_f32(xb, poff[e]);
        }
    };
    auto commit = [&](int st, int c, int tap0, float4 (&wv)[WR], float (&xv)[XR]) {
        if (tap0 == 0) {
            float* xs = xs0 + (c & 1) * XS_ELEMS + ptid;
            // the padding zeros stay zeros under the fused pre-activations used on the hot path (none, leaky ReLU)
            if (p.pre_act == AICG_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < XR; ++e) xs[e * PNT] = xv[e];
            } else if (p.pre_act == AICG_ACT_LRELU) {
#pragma unroll
                for (int e = 0; e < XR; ++e) xs[e * PNT] = xv[e] > 0.f ? xv[e] : xv[e] * p.pre_slope;
            } else {
#pragma unroll
                for (int e = 0; e < XR; ++e) xs[e * PNT] = apply_act(xv[e], p.pre_act, p.pre_slope);
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
    // One register set is enough: after the barrier that publishes stage st - 1 the producers write stage st to the other LDS
    // buffer (its loads were issued one iteration = one consumer stage ago), re-issue the same registers as the loads of stage
    // st + 1 and park at the next barrier until the consumers finish stage st - 1.  (The r1 kernel kept two sets; with every
    // load of a stage now issued as one batch, two sets spilled to scratch inside this loop under the 128-register cap.)
    float4 wv[WR];
    float xv[XR];
    int c = 0, t = 0;
    load(c, t, wv, xv);
#ifdef AICG_CONV_TRACE
    const int ptrace_wg = (ptid < 64 && blockIdx.y == 0 && blockIdx.z == 0) ? (int)blockIdx.x : (1 << 30);
#else
    const int ptrace_wg = 0;
#endif
    for (int st = 0; st < nstages; ++st) {
        trace2(ptrace_wg, 1, st, 0);  // before commit (its first LDS store waits for this stage's loads)
#ifdef AICG_CONV_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trace2(ptrace_wg, 1, st, 1);  // loads landed
#endif
        commit(st, c, t, wv, xv);
        next(c, t);
        if (st + 1 < nstages) load(c, t, wv, xv);
        trace2(ptrace_wg, 1, st, 2);  // arrival at the barrier
        lds_barrier();  // stage st published (and the consumers are done with stage st - 1)
        trace2(ptrace_wg, 1, st, 3);  // released
    }
}

// ---- shared by the 32x32x2 wave-specialised kernels: accumulator start values and the output epilogue ---------------------------
// Lane (l31, half) of a consumer wave owns, per 32 x 32 tile (i, j), rows m0 + (r & 3) + 8 (r >> 2) + 4 half of column nl0 + 32 j + l31.
template <int TM, int TN>
__device__ __forceinline__ void ws_init_acc32(const ConvArgs& p, f32x16 (&acc)[TM][TN], int g, int m_wave0, int half) {
    // The accumulators start from the bias.  A wave's rows are contiguous: lane l loads bias[m_wave0 + 64 t + l] (one or two
    // coalesced loads per wave) and every (tile, register) slot picks its row's value up with a lane shuffle.  (r1: TM x 16
    // serialised global loads; r2 first cut: the same loads batched -- still 16.7k cycles of a 96-row tile's prologue, the
    // memory pipe charges per instruction.)
    if (p.bias) {
        const float* bp = p.bias + g * p.Cout_g;
        const int lane = (int)(threadIdx.x & 63);
        constexpr int NB = (TM * 32 + 63) / 64;
        float breg[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int m = m_wave0 + 64 * t + lane;
            const float v = bp[m < p.Cout_g ? m : 0];
            breg[t] = m < p.Cout_g ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int src = (i & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float b = __shfl(breg[i >> 1], src, 64);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] = b;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
}

// y = [y +] out_scale * (act(acc [+ res]) [+ res | * res]); interior tiles take the predicate-free path.
// The activation is a template parameter of the epilogue body (dispatched once per workgroup): with the run-time switch
// inside, every one of the TM x TN x 16 results per lane walked a chain of scalar branches.
template <int TM, int TN, bool GEN>
__device__ __forceinline__ void ws_epilogue32(const ConvArgs& p, f32x16 (&acc)[TM][TN], int n, int g, int m_wave0, int nl0, int h0, int w0,
                                              int l31, int half, bool interior) {
    const long y_base = (long)n * p.y_sn, r_base = (long)n * p.r_sn;
    auto epilogue = [&](auto full_tag, auto act_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int ACT = decltype(act_tag)::value;
        if constexpr (GEN && FULL) {
            if (p.wide_ok == 2) {
                // kernel = stride = 2 ConvTranspose2d scatter (shuffle == 2), interior tile, 16-byte-aligned rows.  The four registers
                // 4 q .. 4 q + 3 of a tile are the taps (dy, dx) of ONE output channel at this lane's position: rows 2 ho + dy,
                // columns 2 wo + dx.  Lanes (wo, wo + 1) trade halves -- the even lane keeps row dy = 0, the odd one row dy = 1 -- so
                // that each stores FOUR consecutive outputs: one float4 store (+ one float4 skip load) per channel and lane instead of
                // four scattered dwords (+ four loads); a wave writes 16 x 512 B runs (r3: these layers ran at 46-60 TFLOP/s, twice
                // their HBM time, bound by the number of memory instructions).
                const int odd = l31 & 1;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nl = nl0 + j * 32 + l31;
                    const int ho = h0 + (nl >> p.TWlog2), woe = w0 + (nl & (p.TW - 1)) - odd;
                    const long y_pos = y_base + (long)(2 * ho + odd) * p.y_sh + 2 * woe;
                    const long r_pos = r_base + (long)(2 * ho + odd) * p.r_sh + 2 * woe;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int c0 = (g * p.Cout_g + m_wave0 + i * 32 + 4 * half) >> 2;   // output channel of registers 0 .. 3
                        float4 rv[4];
                        if (p.res) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const float4*>(p.res + r_pos + (long)(c0 + 2 * q) * p.r_sc);
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float a0 = acc[i][j][4 * q], a1 = acc[i][j][4 * q + 1], a2 = acc[i][j][4 * q + 2], a3 = acc[i][j][4 * q + 3];
                            const float g0 = __shfl_xor(odd ? a0 : a2, 1, 64), g1 = __shfl_xor(odd ? a1 : a3, 1, 64);
                            float e[4] = {odd ? g0 : a0, odd ? g1 : a1, odd ? a2 : g0, odd ? a3 : g1};
                            const float rr[4] = {p.res ? rv[q].x : 0.f, p.res ? rv[q].y : 0.f, p.res ? rv[q].z : 0.f, p.res ? rv[q].w : 0.f};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                float v = act_static<ACT>(e[t], p.act, p.act_slope);
                                v = p.res_mul ? v * rr[t] : v + rr[t];
                                e[t] = v * p.out_scale;
                            }
                            *reinterpret_cast<float4*>(p.y + y_pos + (long)(c0 + 2 * q) * p.y_sc) = make_float4(e[0], e[1], e[2], e[3]);
                        }
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = nl0 + j * 32 + l31;
            const int ho = h0 + (nl >> p.TWlog2), wo = w0 + (nl & (p.TW - 1));
            const bool col_ok = FULL || (ho < p.Ho && wo < p.Wo);
            const long y_col = y_base + (long)ho * p.y_sh + wo, r_col = r_base + (long)ho * p.r_sh + wo;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m0 = m_wave0 + i * 32 + 4 * half;
                float rv[16], yv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { rv[r] = 0.f; yv[r] = 0.f; }
                if (p.res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (r & 3) + 8 * (r >> 2);
                        const bool ok = FULL || (col_ok && m < p.Cout_g);
                        const float t = p.res[ok ? (GEN ? r_base + out_index(p, g * p.Cout_g + m, ho, wo, p.r_sc, p.r_sh) : r_col + (long)(g * p.Cout_g + m) * p.r_sc) : 0];
                        rv[r] = ok ? t : 0.f;
                    }
                }
                if (p.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (r & 3) + 8 * (r >> 2);
                        const bool ok = FULL || (col_ok && m < p.Cout_g);
                        const float t = p.y[ok ? (GEN ? y_base + out_index(p, g * p.Cout_g + m, ho, wo, p.y_sc, p.y_sh) : y_col + (long)(g * p.Cout_g + m) * p.y_sc) : 0];
                        yv[r] = ok ? t : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (r & 3) + 8 * (r >> 2);
                    if (!FULL && !(col_ok && m < p.Cout_g)) continue;
                    float v = acc[i][j][r];
                    if (p.res_first) v += rv[r];
                    v = act_static<ACT>(v, p.act, p.act_slope);
                    if (!p.res_first) v = (GEN && p.res_mul) ? v * rv[r] : v + rv[r];
                    v = v * p.out_scale + yv[r];
                    if (GEN) p.y[y_base + out_index(p, g * p.Cout_g + m, ho, wo, p.y_sc, p.y_sh)] = v;
                    else p.y[y_col + (long)(g * p.Cout_g + m) * p.y_sc] = v;
                }
            }
        }
    };
    dispatch_epilogue(p.act, interior, epilogue);
}


// Interior tiles, plain output addressing, 16-byte-aligned rows: the float4 epilogue.
// In the MFMA accumulator layout a lane owns ONE position and 16 rows per 32 x 32 tile: a tile costs 16 dword stores per lane (plus
// 16 + 16 dword loads with a residual / accumulate operand), and the tail of a tile is bound by the NUMBER of global memory
// instructions, not by bytes (r2 timeline: 38.5k cycles for the 48 stores of a 96 x 128 tile's wave, ~800 cycles per instruction
// -- cf. MI355X guide T21).  Each 32 x 32 tile therefore takes a detour through a per-wave LDS scratch (16 ds_write_b32, 4
// ds_read_b128) that turns the layout into 4 consecutive positions per lane: 4 global_store_dwordx4 per tile, 8 rows x 128 B each.
static constexpr int kEpiRow = 36;                 // scratch row stride in floats (32 + 4: keeps float4 alignment, spreads banks)
static constexpr int kEpiScratch = 32 * kEpiRow;   // floats per consumer wave
template <int TM, int TN>
__device__ __forceinline__ void ws_epilogue32_wide(const ConvArgs& p, f32x16 (&acc)[TM][TN], int n, int g, int m_wave0, int nl0, int h0,
                                                   int w0, int lane, float* scratch) {
    const int l31 = lane & 31, half = lane >> 5;
    const int rrow = lane >> 3, rcol = (lane & 7) * 4;   // after the detour: rows rrow + 8 pass, positions rcol .. rcol + 3
    float* wr = scratch + (4 * half) * kEpiRow + l31;
    const float4* rd = reinterpret_cast<const float4*>(scratch + rrow * kEpiRow + rcol);
    auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = nl0 + j * 32 + rcol;
            const int ho = h0 + (nl >> p.TWlog2), wo = w0 + (nl & (p.TW - 1));
            const long y_col = (long)n * p.y_sn + (long)ho * p.y_sh + wo, r_col = (long)n * p.r_sn + (long)ho * p.r_sh + wo;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                __builtin_amdgcn_wave_barrier();   // (scheduling fence; the emulator's lanes rendezvous: previous tile fully read)
#pragma unroll
                for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * kEpiRow] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();   // all lanes' rows written before any lane reads across them
                const long co0 = (long)g * p.Cout_g + m_wave0 + i * 32 + rrow;
                float4 rv[4], yv[4], v[4];
                if (p.res) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const float4*>(p.res + r_col + (co0 + 8 * q) * p.r_sc);
                }
                if (p.accumulate) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) yv[q] = *reinterpret_cast<const float4*>(p.y + y_col + (co0 + 8 * q) * p.y_sc);
                }
                // (a wave's LDS accesses are processed in order: no s_barrier between its own writes and reads)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = rd[q * 8 * (kEpiRow / 4)];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
                    const float rr[4] = {p.res ? rv[q].x : 0.f, p.res ? rv[q].y : 0.f, p.res ? rv[q].z : 0.f, p.res ? rv[q].w : 0.f};
                    const float yy[4] = {p.accumulate ? yv[q].x : 0.f, p.accumulate ? yv[q].y : 0.f, p.accumulate ? yv[q].z : 0.f,
                                         p.accumulate ? yv[q].w : 0.f};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float x = e[t];
                        if (p.res_first) x += rr[t];
                        x = act_static<ACT>(x, p.act, p.act_slope);
                        if (!p.res_first) x += rr[t];
                        e[t] = x * p.out_scale + yy[t];
                    }
                    *reinterpret_cast<float4*>(p.y + y_col + (co0 + 8 * q) * p.y_sc) = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
        }
    };
    if (p.act == AICG_ACT_NONE) body(std::integral_constant<int, 0>{});
    else if (p.act == AICG_ACT_RELU) body(std::integral_constant<int, 1>{});
    else if (p.act == AICG_ACT_LRELU) body(std::integral_constant<int, 2>{});
    else body(std::integral_constant<int, 3>{});
}

// host side: may interior tiles of this launch use ws_epilogue32_wide?
inline int conv_wide_ok(const ConvArgs& p) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    auto m4 = [](long v) { return (v & 3) == 0; };
    if (p.shuffle == 2 && !p.accumulate && !p.res_first && (p.TW & 1) == 0 && (p.Cout_g & 3) == 0 && (!p.res_mul || p.res) && al(p.y) &&
        m4(p.y_sn) && m4(p.y_sc) && m4(p.y_sh) && (!p.res || (al(p.res) && m4(p.r_sn) && m4(p.r_sc) && m4(p.r_sh))))
        return 2;   // the paired-lane float4 scatter of ws_epilogue32<.., GEN = true>
    if (p.shuffle || p.res_mul || (p.TW & 3)) return 0;
    if (!al(p.y) || !m4(p.y_sn) || !m4(p.y_sc) || !m4(p.y_sh)) return 0;
    if (p.res && (!al(p.res) || !m4(p.r_sn) || !m4(p.r_sc) || !m4(p.r_sh))) return 0;
    return 1;
}

// GEN: instantiation for shuffle / multiplicative-residual layers (runtime-generic output addressing); kept out of the plain
// instantiation, whose register allocation it would disturb.
template <int BM, int BN, int WM, int WN, int XR, int KS, bool GEN>
__global__ void __launch_bounds__(64 * (WM * WN + 4), (WM * WN == 4 ? 4 : 3)) conv_ws_kernel(ConvArgs p) {
    constexpr int CW = WM * WN, CNT = 64 * CW, PNT = 256;
    constexpr int TM = BM / (32 * WM);
    constexpr int TN = BN / (32 * WN);
    constexpr int WR = WsGeom<BM, KS>::WR;
    constexpr int WS_ELEMS = WsGeom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = XR * PNT;
    HIP_DYNAMIC_SHARED(float, smem)
    float* const xs0 = smem;
    float* const ws0 = smem + 2 * XS_ELEMS;

    const int tid = threadIdx.x;
    const int bx = (kAblate && (p.dbg & 32)) ? (int)blockIdx.x : (int)xcd_remap(blockIdx.x, gridDim.x);
    const int tw_i = bx % p.tiles_w;
    const int th_i = (bx / p.tiles_w) % p.tiles_h;
    const int n = bx / (p.tiles_w * p.tiles_h);
    const int w0 = tw_i * p.TW, h0 = th_i * p.TH;
    const int m_base = blockIdx.y * BM;
    const int g = blockIdx.z;
    const int stages_per_chunk = (p.taps + p.TT - 1) / p.TT;
    const int nstages = p.nchunk * stages_per_chunk;

    if (tid >= CNT) {
        ws_produce<BM, XR, KS, CW == 8>(p, xs0, ws0, tid - CNT, n, g, h0, w0, m_base, nstages);
        return;
    }

    // ================= consumers =================
    const int lane = tid & 63, wave = tid >> 6;
#ifdef AICG_CONV_TRACE
    const int trace_wg = (wave == 0 && blockIdx.y == 0 && blockIdx.z == 0) ? (int)blockIdx.x : (1 << 30);
    trace_mark(trace_wg, 0);
    trace_val(trace_wg, 5, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));   // HW_REG_HW_ID
#else
    const int trace_wg = 0;
#endif
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (TN * 32) + j * 32 + l31;
        const int jh = nl >> p.TWlog2, jw = nl & (p.TW - 1);
        boff[j] = jh * p.sh * p.TWp + jw * p.sw + half * p.CHS;
    }
    trace_mark(trace_wg, 6);
    if (CW == 4 && p.stagger > 0) {
        const unsigned flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        // HW_REG_HW_ID[3:0] = wave slot on the SIMD: this workgroup has 2 waves per SIMD, so slots >= 2 mean another
        // workgroup was placed on the CU first
        if (flat < (unsigned)p.stagger_first && (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 0xfu) >= 2u) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(64);
        }
    }
    f32x16 acc[TM][TN];
    ws_init_acc32<TM, TN>(p, acc, g, m_base + wm * (TM * 32), half);
    const int a_off = wm * (TM * 32) + l31 + half * BM;
#ifdef AICG_CONV_TRACE
    if (acc[0][0][0] == 1.2345e-30f) return;   // consume the bias loads before the stamp
#endif
    trace_mark(trace_wg, 7);
    {
        int c = 0, tap0 = 0;
        for (int st = 0; st < nstages; ++st) {
            trace2(trace_wg, 0, st, 0);   // arrival at the barrier (done with stage st - 1)
            lds_barrier();  // stage st is in LDS
            trace2(trace_wg, 0, st, 1);   // released
            if (st == 0) trace_mark(trace_wg, 1);
            if (kAblate && (p.dbg & 8)) continue;
            const float* xs = xs0 + (c & 1) * XS_ELEMS;
            const float* wt = ws0 + (st & 1) * WS_ELEMS + a_off;
            const int nt = imin(p.TT, p.taps - tap0);
            const int nsteps = nt * (p.BKC >> 1);
            float a0[TM], b0[TN], a1[TM], b1[TN];
            // patch offset of the (tap, channel pair) being fetched, advanced incrementally: + 2 channels per k-step, then to
            // the next column tap, then to the next kernel row
            const int kh0 = tap0 / p.KW;
            int kw = tap0 - kh0 * p.KW, kk = 0;
            int xoff = kh0 * p.dh * p.TWp + kw * p.dw;
            const int step_k = 2 * p.CHS, next_tap = p.dw - p.BKC * p.CHS, next_row = p.dh * p.TWp - p.KW * p.dw;
            auto fetch = [&](float (&a)[TM], float (&b)[TN], int s) {
                const float* xt = xs + xoff;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wt[s * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = xt[boff[j]];
                xoff += step_k;
                kk += 2;
                if (kk == p.BKC) { kk = 0; xoff += next_tap; if (++kw == p.KW) { kw = 0; xoff += next_row; } }
            };
            auto mma = [&](float (&a)[TM], float (&b)[TN]) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            };
            fetch(a0, b0, 0);
            int s = 0;
            for (; s + 2 <= nsteps; s += 2) {
                fetch(a1, b1, s + 1);
                mma(a0, b0);
                fetch(a0, b0, s + 2);
                mma(a1, b1);
            }
            if (s < nsteps) mma(a0, b0);
            tap0 += p.TT;
            if (tap0 >= p.taps) { tap0 = 0; ++c; }
        }
    }
    trace_mark(trace_wg, 2);
    if (kAblate && (p.dbg & 16)) { if (acc[0][0][0] != 12345.f) return; }

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

// ---- shared by the 16x16x4 wave-specialised kernels -----------------------------------------------------------------------------
// Lane (r16, q) owns, per 16 x 16 tile (i, j), rows m_base + 16 i + 4 q + r (r = 0..3) of column nl0 + 16 j + r16.
template <int TM, int TN>
__device__ __forceinline__ void ws_init_acc16(const ConvArgs& p, f32x4 (&acc)[TM][TN], int g, int m_base, int q) {
    if (p.bias) {  // one coalesced load of the tile's rows per wave, then a lane shuffle per (tile, register) slot
        const float* bp = p.bias + g * p.Cout_g;
        const int lane = (int)(threadIdx.x & 63);
        const int mb = m_base + lane;
        const float bl = bp[(lane < TM * 16 && mb < p.Cout_g) ? mb : 0];
        const float breg = (lane < TM * 16 && mb < p.Cout_g) ? bl : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b = __shfl(breg, i * 16 + q * 4 + r, 64);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] = b;
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    }
}

template <int TM, int TN, bool GEN>
__device__ __forceinline__ void ws_epilogue16(const ConvArgs& p, f32x4 (&acc)[TM][TN], int n, int g, int m_base, int nl0, int h0, int w0,
                                              int r16, int q, bool interior) {
    const long y_base = (long)n * p.y_sn, r_base = (long)n * p.r_sn;
    auto epilogue = [&](auto full_tag, auto act_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = nl0 + j * 16 + r16;
            const int ho = h0 + (nl >> p.TWlog2), wo = w0 + (nl & (p.TW - 1));
            const bool col_ok = FULL || (ho < p.Ho && wo < p.Wo);
            const long y_col = y_base + (long)ho * p.y_sh + wo, r_col = r_base + (long)ho * p.r_sh + wo;
            float rv[TM * 4], yv[TM * 4];
#pragma unroll
            for (int e = 0; e < TM * 4; ++e) { rv[e] = 0.f; yv[e] = 0.f; }
            if (p.res) {
#pragma unroll
                for (int e = 0; e < TM * 4; ++e) {
                    const int m = m_base + (e >> 2) * 16 + q * 4 + (e & 3);
                    const bool ok = FULL || (col_ok && m < p.Cout_g);
                    const float t = p.res[ok ? (GEN ? r_base + out_index(p, g * p.Cout_g + m, ho, wo, p.r_sc, p.r_sh) : r_col + (long)(g * p.Cout_g + m) * p.r_sc) : 0];
                    rv[e] = ok ? t : 0.f;
                }
            }
            if (p.accumulate) {
#pragma unroll
                for (int e = 0; e < TM * 4; ++e) {
                    const int m = m_base + (e >> 2) * 16 + q * 4 + (e & 3);
                    const bool ok = FULL || (col_ok && m < p.Cout_g);
                    const float t = p.y[ok ? (GEN ? y_base + out_index(p, g * p.Cout_g + m, ho, wo, p.y_sc, p.y_sh) : y_col + (long)(g * p.Cout_g + m) * p.y_sc) : 0];
                    yv[e] = ok ? t : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < TM * 4; ++e) {
                const int m = m_base + (e >> 2) * 16 + q * 4 + (e & 3);
                if (!FULL && !(col_ok && m < p.Cout_g)) continue;
                float v = acc[e >> 2][j][e & 3];
                if (p.res_first) v += rv[e];
                v = act_static<ACT>(v, p.act, p.act_slope);
                if (!p.res_first) v = (GEN && p.res_mul) ? v * rv[e] : v + rv[e];
                v = v * p.out_scale + yv[e];
                if (GEN) p.y[y_base + out_index(p, g * p.Cout_g + m, ho, wo, p.y_sc, p.y_sh)] = v;
                else p.y[y_col + (long)(g * p.Cout_g + m) * p.y_sc] = v;
            }
        }
    };
    dispatch_epilogue(p.act, interior, epilogue);
}

// float4 epilogue of the 16x16x4 kernels (interior tiles, plain addressing, aligned rows): per 16-row block a wave's 16 x 64 outputs
// take the detour through a per-wave LDS scratch (16 ds_write_b32 + 4 ds_read_b128) and leave as 4 global_store_dwordx4 covering
// 4 rows x 256 B each, instead of 16 dword stores per lane (see ws_epilogue32_wide).
static constexpr int kEpi16Row = 68;                  // 64 positions + 4
static constexpr int kEpi16Scratch = 16 * kEpi16Row;  // floats per consumer wave
template <int TM, int TN>
__device__ __forceinline__ void ws_epilogue16_wide(const ConvArgs& p, f32x4 (&acc)[TM][TN], int n, int g, int m_base, int nl0, int h0, int w0,
                                                   int lane, float* scratch) {
    static_assert(TN == 4, "a consumer wave covers 64 positions");
    const int r16 = lane & 15, q = lane >> 4;
    const int rrow = lane >> 4, rcol = (lane & 15) * 4;   // after the detour: rows rrow + 4 pass, positions rcol .. rcol + 3
    float* wr = scratch + (4 * q) * kEpi16Row + r16;
    const float4* rd = reinterpret_cast<const float4*>(scratch + rrow * kEpi16Row + rcol);
    const int nl = nl0 + rcol;
    const int ho = h0 + (nl >> p.TWlog2), wo = w0 + (nl & (p.TW - 1));
    const long y_col = (long)n * p.y_sn + (long)ho * p.y_sh + wo, r_col = (long)n * p.r_sn + (long)ho * p.r_sh + wo;
    auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) wr[r * kEpi16Row + j * 16] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            const long co0 = (long)g * p.Cout_g + m_base + i * 16 + rrow;
            float4 rv[4], yv[4], v[4];
            if (p.res) {
#pragma unroll
                for (int t = 0; t < 4; ++t) rv[t] = *reinterpret_cast<const float4*>(p.res + r_col + (co0 + 4 * t) * p.r_sc);
            }
            if (p.accumulate) {
#pragma unroll
                for (int t = 0; t < 4; ++t) yv[t] = *reinterpret_cast<const float4*>(p.y + y_col + (co0 + 4 * t) * p.y_sc);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = rd[t * 4 * (kEpi16Row / 4)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float e[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
                const float rr[4] = {p.res ? rv[t].x : 0.f, p.res ? rv[t].y : 0.f, p.res ? rv[t].z : 0.f, p.res ? rv[t].w : 0.f};
                const float yy[4] = {p.accumulate ? yv[t].x : 0.f, p.accumulate ? yv[t].y : 0.f, p.accumulate ? yv[t].z : 0.f,
                                     p.accumulate ? yv[t].w : 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float x = e[u];
                    if (p.res_first) x += rr[u];
                    x = act_static<ACT>(x, p.act, p.act_slope);
                    if (!p.res_first) x += rr[u];
                    e[u] = x * p.out_scale + yy[u];
                }
                *reinterpret_cast<float4*>(p.y + y_col + (co0 + 4 * t) * p.y_sc) = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    if (p.act == AICG_ACT_NONE) body(std::integral_constant<int, 0>{});
    else if (p.act == AICG_ACT_RELU) body(std::integral_constant<int, 1>{});
    else if (p.act == AICG_ACT_LRELU) body(std::integral_constant<int, 2>{});
    else body(std::integral_constant<int, 3>{});
}

// Wave-specialised narrow-M kernel: the consumers of conv_mfma16_kernel (16x16x4 MFMA, every wave covers all BM rows x 64
// positions of a 256-position tile) fed by ws_produce.  WsGeom pads the weight stage with 4 slack rows (one 16x16x4 k-step).
template <int BM, int XR, int KS, bool GEN>
__global__ void __launch_bounds__(512, 4) conv_ws16_kernel(ConvArgs p) {
    constexpr int CNT = 256;
    constexpr int TM = BM / 16, TN = 4;
    constexpr int WS_ELEMS = WsGeom<BM, KS>::WS_ELEMS;
    constexpr int XS_ELEMS = XR * 256;
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
        ws_produce<BM, XR, KS, false>(p, xs0, ws0, tid - CNT, n, g, h0, w0, m_base, nstages);
        return;
    }
    const int lane = tid & 63, wn = tid >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * 64 + j * 16 + r16;
        const int jh = nl >> p.TWlog2, jw = nl & (p.TW - 1);
        boff[j] = jh * p.sh * p.TWp + jw * p.sw + q * p.CHS;
    }
    f32x4 acc[TM][TN];
    ws_init_acc16<TM, TN>(p, acc, g, m_base, q);
    {
        int c = 0, tap0 = 0;
        for (int st = 0; st < nstages; ++st) {
            lds_barrier();  // stage st is in LDS
            const float* xs = xs0 + (c & 1) * XS_ELEMS;
            const float* wt = ws0 + (st & 1) * WS_ELEMS + q * BM + r16;
            const int nt = imin(p.TT, p.taps - tap0);
            const int nsteps = nt * (p.BKC >> 2);  // k-steps of 4 rows
            float a0[TM], b0[TN], a1[TM], b1[TN];
            const int kh0 = tap0 / p.KW;
            int kw = tap0 - kh0 * p.KW, kk = 0;
            int xoff = kh0 * p.dh * p.TWp + kw * p.dw;
            const int step_k = 4 * p.CHS, next_tap = p.dw - p.BKC * p.CHS, next_row = p.dh * p.TWp - p.KW * p.dw;
            auto fetch = [&](float (&a)[TM], float (&b)[TN], int s) {
                const float* xt = xs + xoff;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = wt[s * 4 * BM + i * 16];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = xt[boff[j]];
                xoff += step_k;
                kk += 4;
                if (kk == p.BKC) { kk = 0; xoff += next_tap; if (++kw == p.KW) { kw = 0; xoff += next_row; } }
            };
            auto mma = [&](float (&a)[TM], float (&b)[TN]) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            };
            fetch(a0, b0, 0);
            int s = 0;
            for (; s + 2 <= nsteps; s += 2) {
                fetch(a1, b1, s + 1);
                mma(a0, b0);
                fetch(a0, b0, s + 2);
                mma(a1, b1);
            }
            if (s < nsteps) mma(a0, b0);
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

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline unsigned div_mul(int d) { return (unsigned)((0x100000000ULL + (unsigned long long)d - 1) / (unsigned long long)d); }

template <int BM, int BN, int WM, int WN, int XR>
static int launch_conv_xr(ConvArgs& p, hipStream_t stream, size_t lds) {
    auto kern = conv_mfma_kernel<BM, BN, WM, WN, XR>;
    if (lds > 64 * 1024)
        allow_dynamic_lds((const void*)kern, lds);
    const long gx = (long)p.N * p.tiles_h * p.tiles_w;
    if (gx > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    dim3 grid((unsigned)gx, (unsigned)idiv_up(p.Cout_g, BM), (unsigned)p.groups);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, stream, p);
    return check_launch("conv_mfma_kernel");
}

// Output tile = TH x TW positions with TH*TW = BN.  A flat 1 x BN tile re-stages (KH-1) halo rows per output row; for 2-D
// layers pick the power-of-two TW (>= 16 for coalesced patch rows) that minimises staged input elements over the layer.
inline int choose_tile_width(const ConvArgs& p, int BN) {
    int TW = 1 << ilog2(p.Wo);
    if (TW > BN) TW = BN;
    if (p.Ho == 1) return BN;
    AICG_SWITCH(square, "AICG_CONV_SQUARE", 1);
    if (!square) return TW;
    int best = TW;
    long best_cost = -1;
    for (int tw = TW; tw >= 16; tw >>= 1) {
        const int th = BN / tw;
        const int th_in = (th - 1) * p.sh + (p.KH - 1) * p.dh + 1;
        const int tw_in = (tw - 1) * p.sw + (p.KW - 1) * p.dw + 1;
        const long cost = (long)th_in * (tw_in | 1) * idiv_up(p.Wo, tw) * idiv_up(p.Ho, th);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = tw; }
    }
    return best;
}

template <int BM, int BN, int WM, int WN>
static int launch_conv(ConvArgs& p, hipStream_t stream) {
    // output patch: TW (power of two) columns x TH rows = BN positions
    p.TW = choose_tile_width(p, BN);
    p.TWlog2 = ilog2(p.TW);
    p.TH = BN / p.TW;
    p.TH_in = (p.TH - 1) * p.sh + (p.KH - 1) * p.dh + 1;
    p.TW_in = (p.TW - 1) * p.sw + (p.KW - 1) * p.dw + 1;
    p.TWp = p.TW_in | 1;
    p.CHS = p.TH_in * p.TWp;
    p.tiles_w = idiv_up(p.Wo, p.TW);
    p.tiles_h = idiv_up(p.Ho, p.TH);
    // channels per K chunk: as many as keep the staged patch within 8 prefetch registers per thread (2048 floats)
    // and a weight stage within KSTAGE rows; never (much) more than the layer has
    p.BKC = 32;
    constexpr int XRMAX = (WM * WN == 8) ? 8 : 12;  // 8-wave tiles run at 4 waves/SIMD: 128 registers per lane
    while (p.BKC > 2 && (p.BKC * p.CHS > XRMAX * 64 * WM * WN || p.BKC >= 2 * p.Cin_g)) p.BKC >>= 1;
    p.BKClog2 = ilog2(p.BKC);
    {   // taps per weight stage: as few, equally sized stages per chunk as fit KSTAGE rows
        const int cap = imax(1, KSTAGE / p.BKC);
        const int nstg = idiv_up(p.taps, cap);
        p.TT = idiv_up(p.taps, nstg);
    }
    p.nchunk = idiv_up(p.Cin_g, p.BKC);
    p.xs_total = p.BKC * p.CHS;
    p.xs_elems = (p.xs_total + 3) & ~3;
    p.div_chs = div_mul(p.CHS);
    p.div_twp = div_mul(p.TWp);
    // + 2 weight rows of slack: the MFMA loop's last (discarded) fragment prefetch reads one k-step past the stage
    const size_t lds = (size_t)(p.xs_elems + (KSTAGE + 2) * BM) * sizeof(float);
    const int xr = idiv_up(p.xs_total, 64 * WM * WN);
    if (lds > 160 * 1024 || (long)p.xs_total * p.CHS >= (1L << 32))
        return fail(AICG_E_LDS, "conv: input patch of %d x %d x %d floats is too large for one workgroup (stride/kernel too big: "
                                "re-express the layer with the phase decomposition used for Cin = 1 convs)", p.BKC, p.TH_in, p.TWp);
    if (xr <= 8) return launch_conv_xr<BM, BN, WM, WN, 8>(p, stream, lds);
    if (xr <= 12) return launch_conv_xr<BM, BN, WM, WN, 12>(p, stream, lds);
    return fail(AICG_E_LDS, "conv: a 2-channel input patch of %d floats exceeds the staging budget", p.xs_total);
}

// wave-specialised launch: returns 1 when the configuration does not fit (caller uses conv_mfma_kernel)
template <int BM, int BN, int WM, int WN, int KS>
static int launch_conv_ws(ConvArgs& p, hipStream_t stream) {
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
    while (p.BKC > 2 && (p.BKC * p.CHS > 12 * 256 || p.BKC >= 2 * p.Cin_g)) p.BKC >>= 1;
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
    const int xr = idiv_up(p.xs_total, 256) <= 8 ? 8 : 12;
    const size_t lds = (size_t)(2 * xr * 256 + 2 * WsGeom<BM, KS>::WS_ELEMS) * sizeof(float);
    // hoisted producer offsets are 32-bit byte offsets below kBufOob: one channel chunk of the input / one group of packed
    // weights must span < 2^31 bytes
    const bool off_ok = (long)p.BKC * p.x_sc + (long)p.H * p.x_sh < (1L << 29) && (long)p.taps * p.Cin_pad * p.Mpad < (1L << 29);
    if (lds > 160 * 1024 || p.xs_total > 12 * 256 || !off_ok || (long)p.xs_total * p.CHS >= (1L << 32)) return 1;
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
            const long ksteps = (long)p.nchunk * p.taps * (p.BKC / 2);
            const long floor_cycles = ksteps * (BM / (32 * WM)) * (BN / (32 * WN)) * 64;   // MFMA issue time of one tile
            p.stagger = (int)lmin(floor_cycles + 16000, 1L << 22) * (stag > 1 ? stag : 1) / (stag > 1 ? 100 : 1);
            p.stagger_first = 512;
        }
    }
#endif
    auto kern = gen ? (xr == 8 ? conv_ws_kernel<BM, BN, WM, WN, 8, KS, true> : conv_ws_kernel<BM, BN, WM, WN, 12, KS, true>)
                    : (xr == 8 ? conv_ws_kernel<BM, BN, WM, WN, 8, KS, false> : conv_ws_kernel<BM, BN, WM, WN, 12, KS, false>);
    allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return check_launch("conv_ws_kernel");
}

template <int BM>
static int launch_conv16(ConvArgs& p, hipStream_t stream) {
    constexpr int BN = 256;
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
    while (p.BKC > 4 && (p.BKC * p.CHS > 12 * 256 || p.BKC >= 2 * p.Cin_g)) p.BKC >>= 1;
    p.BKClog2 = ilog2(p.BKC);
    {
        const int cap = imax(1, KSTAGE / p.BKC);
        const int nstg = idiv_up(p.taps, cap);
        p.TT = idiv_up(p.taps, nstg);
    }
    p.nchunk = idiv_up(p.Cin_g, p.BKC);
    p.xs_total = p.BKC * p.CHS;
    p.xs_elems = (p.xs_total + 3) & ~3;
    p.div_chs = div_mul(p.CHS);
    p.div_twp = div_mul(p.TWp);
    const size_t lds = (size_t)(p.xs_elems + (KSTAGE + 4) * BM) * sizeof(float);
    const int xr = idiv_up(p.xs_total, 256);
    if (lds > 160 * 1024 || xr > 12 || (long)p.xs_total * p.CHS >= (1L << 32)) return 1;  // caller falls back to the 32x32 kernel
    const long gx = (long)p.N * p.tiles_h * p.tiles_w;
    if (gx > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    dim3 grid((unsigned)gx, (unsigned)idiv_up(p.Cout_g, BM), (unsigned)p.groups);
    AICG_SWITCH(ws, "AICG_CONV_WS", 1);
    const bool off_ok = (long)p.BKC * p.x_sc + (long)p.H * p.x_sh < (1L << 29) && (long)p.taps * p.Cin_pad * p.Mpad < (1L << 29);
    if (ws && off_ok) {  // wave-specialised form (KSTAGE rows per stage, double-buffered)
        const int xrw = xr <= 8 ? 8 : 12;
        const size_t ldsw = (size_t)(2 * xrw * 256 + 2 * WsGeom<BM, KSTAGE>::WS_ELEMS) * sizeof(float);
        const bool gen = p.shuffle || p.res_mul;
        {
            AICG_SWITCH(wide, "AICG_CONV_WIDE", 1);
            p.wide_ok = wide ? conv_wide_ok(p) : 0;
        }
        auto kern = gen ? (xrw == 8 ? conv_ws16_kernel<BM, 8, KSTAGE, true> : conv_ws16_kernel<BM, 12, KSTAGE, true>)
                        : (xrw == 8 ? conv_ws16_kernel<BM, 8, KSTAGE, false> : conv_ws16_kernel<BM, 12, KSTAGE, false>);
        allow_dynamic_lds((const void*)kern, ldsw);
        hipLaunchKernelGGL(kern, grid, dim3(512), ldsw, stream, p);
        return check_launch("conv_ws16_kernel");
    }
    if (xr <= 8) {
        auto kern = conv_mfma16_kernel<BM, 8>;
        allow_dynamic_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    } else {
        auto kern = conv_mfma16_kernel<BM, 12>;
        allow_dynamic_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    }
    return check_launch("conv_mfma16_kernel");
}



// ---- instantiation units (each defined in its own .hip file so that hipcc compiles them in parallel) ----------------------
// wave-specialised 32x32x2 tiles: return 0 launched, < 0 error, 1 the configuration does not fit (use a single-role kernel)
int run_ws_160x128(ConvArgs& p, hipStream_t st);    // conv_ws_1.hip
int run_ws_128x128(ConvArgs& p, hipStream_t st);
int run_ws_96x128(ConvArgs& p, hipStream_t st);     // conv_ws_2.hip
int run_ws_64x128(ConvArgs& p, hipStream_t st);
int run_ws_64x64(ConvArgs& p, hipStream_t st);      // conv_ws_3.hip
int run_ws_32x256(ConvArgs& p, hipStream_t st);
int run_ws_32x128(ConvArgs& p, hipStream_t st);
int run_ws_128x128_k32(ConvArgs& p, hipStream_t st);   // conv_ws_4.hip: 4 consumers x (128 x 32), 32-row stages: two workgroups per CU
int run_ws_64x256(ConvArgs& p, hipStream_t st);        //                4 consumers x (64 x 64)
int run_ws_32x512(ConvArgs& p, hipStream_t st);        //                4 consumers x (32 x 128)
// 16x16x4 tiles for 48- / 16-row layers (wave-specialised, or single-role when AICG_CONV_WS=0): 1 = not applicable
int run_m16_48(ConvArgs& p, hipStream_t st);        // conv_ws16.hip
int run_m16_16(ConvArgs& p, hipStream_t st);
// single-role kernels
int run_sr_160x128(ConvArgs& p, hipStream_t st);    // conv_single_role.hip
int run_sr_128x128_8w(ConvArgs& p, hipStream_t st);
int run_sr_128x128_4w(ConvArgs& p, hipStream_t st);
int run_sr_96x128(ConvArgs& p, hipStream_t st);
int run_sr_64x128(ConvArgs& p, hipStream_t st);     // conv_single_role_2.hip
int run_sr_64x64(ConvArgs& p, hipStream_t st);
int run_sr_32x256(ConvArgs& p, hipStream_t st);
int run_sr_32x128(ConvArgs& p, hipStream_t st);

}  // namespace aicg
