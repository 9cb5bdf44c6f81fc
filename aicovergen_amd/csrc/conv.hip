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
#include <type_traits>

#include "common.h"

#include <cstdint>
#include <cstdlib>

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
    int shuffle;   // 2: GEMM row m / position (ho, wo) -> y[m >> 2][2 ho + ((m >> 1) & 1)][2 wo + (m & 1)] (k = s = 2 ConvTranspose2d)
    int res_mul;   // the residual operand multiplies (U-Net multiplicative skip) instead of adding, after the activation
    // derived tiling
    int TW, TWlog2, TH, TH_in, TW_in, TWp, CHS, BKC, BKClog2, TT, tiles_w, tiles_h, nchunk, taps, Mpad, Cin_pad, xs_elems, xs_total;
    unsigned div_chs, div_twp;  // ceil(2^32 / d) multipliers: idx / d == umulhi(idx, mul) for idx * d < 2^32
    long w_group_stride;
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
        if (p.dbg & 1) return;
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
        if (p.dbg & 2) return;
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
        if (!(p.dbg & 4)) __syncthreads();  // every wave has finished the MFMAs of the previous stage: LDS may be overwritten
        commit(st);
        if (!(p.dbg & 4)) __syncthreads();
        if (st + 1 < nstages) prefetch(st + 1);  // in flight during the MFMA loop below
        if (p.dbg & 8) continue;

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
    if (p.dbg & 16) { if (acc[0][0][0] != 12345.f) return; }
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
    if (p.dbg & 1) {
        for (int st = 0; st < nstages; ++st) lds_barrier();
        return;
    }
    // BOOST: one workgroup per CU (8-consumer shape): nothing else runs while its consumers wait for a late stage
    if (BOOST) __builtin_amdgcn_s_setprio(2);
    int poff[XR], woff[WR];
    unsigned pmask = 0, wmask = 0;
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
            poff[e] = ok ? (int)(ci * p.x_sc + hin * p.x_sh + win) : 0;
            pmask |= ok ? (1u << e) : 0u;
        }
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const int idx4 = ptid + e * PNT;
            const int r = idx4 / (BM / 4);
            const int c4 = idx4 - r * (BM / 4);
            const int tt = r >> p.BKClog2, ci = r & (p.BKC - 1);
            const bool ok = m_base + c4 * 4 < p.Mpad && r < KS;
            woff[e] = ok ? (tt * p.Cin_pad + ci) * p.Mpad + c4 * 4 : 0;
            wmask |= ok ? (1u << e) : 0u;
        }
    }
    // Two register sets: the global loads of stage st + 1 are issued before stage st is written to LDS, so a
    // stage's load latency spans a whole consumer stage.  lds_barrier() does not wait for loads in flight.
    auto load = [&](int c, int tap0, float4 (&wv)[WR], float (&xv)[XR]) {
        const int wlim = (imin(p.TT, p.taps - tap0) << p.BKClog2) * (BM / 4);  // idx4 below this: a row of this stage
        const float* wrow0 = wg + ((long)tap0 * p.Cin_pad + (long)c * p.BKC) * p.Mpad;
#pragma unroll
        for (int e = 0; e < WR; ++e) {
            const bool ok = ((wmask >> e) & 1u) && (ptid + e * PNT) < wlim && !(p.dbg & 512);
            const float4 t = *reinterpret_cast<const float4*>(wrow0 + (ok ? woff[e] : 0));
            wv[e] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tap0 == 0) {  // a new channel chunk: its input patch (halo included)
            const float* xc = xg + (long)c * p.BKC * p.x_sc;
            const int xlim = imin(p.BKC, p.Cin_g - c * p.BKC) * p.CHS;  // idx below this: a channel the layer has
#pragma unroll
            for (int e = 0; e < XR; ++e) {
                const bool ok = ((pmask >> e) & 1u) && (ptid + e * PNT) < xlim && !(p.dbg & 256);
                const float t = xc[ok ? poff[e] : 0];
                xv[e] = ok ? t : 0.f;
            }
        }
    };
    auto commit = [&](int st, int c, int tap0, float4 (&wv)[WR], float (&xv)[XR]) {
        if (tap0 == 0) {
            float* xs = xs0 + (c & 1) * XS_ELEMS + ptid;
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
    float4 wvA[WR], wvB[WR];
    float xvA[XR], xvB[XR];
    int cA = 0, tA = 0, cB = 0, tB = 0;
    load(cA, tA, wvA, xvA);
    for (int st = 0; st < nstages; st += 2) {
        cB = cA; tB = tA; next(cB, tB);
        if (st + 1 < nstages) load(cB, tB, wvB, xvB);
        commit(st, cA, tA, wvA, xvA);
        lds_barrier();  // stage st published (and the consumers are done with stage st - 1)
        if (st + 1 < nstages) {
            cA = cB; tA = tB; next(cA, tA);
            if (st + 2 < nstages) load(cA, tA, wvA, xvA);
            commit(st + 1, cB, tB, wvB, xvB);
            lds_barrier();
        }
    }
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
    const int bx = (p.dbg & 32) ? (int)blockIdx.x : (int)xcd_remap(blockIdx.x, gridDim.x);
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
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * (TN * 32) + j * 32 + l31;
        const int jh = nl >> p.TWlog2, jw = nl & (p.TW - 1);
        boff[j] = jh * p.sh * p.TWp + jw * p.sw + half * p.CHS;
    }
    // the accumulators start from the bias (its loads overlap the wait for the first stage)
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m0 = m_base + wm * (TM * 32) + i * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2);
            float b = 0.f;
            if (p.bias) {
                const float t = p.bias[g * p.Cout_g + (m < p.Cout_g ? m : 0)];
                b = m < p.Cout_g ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j][r] = b;
        }
    }
    const int a_off = wm * (TM * 32) + l31 + half * BM;
    {
        int c = 0, tap0 = 0;
        for (int st = 0; st < nstages; ++st) {
            lds_barrier();  // stage st is in LDS
            if (p.dbg & 8) continue;
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
    if (p.dbg & 16) { if (acc[0][0][0] != 12345.f) return; }

    // ---- epilogue: y = [y +] out_scale * (act(acc [+ res]) [+ res]); interior tiles take the predicate-free path ----
    const long y_base = (long)n * p.y_sn, r_base = (long)n * p.r_sn;
    const bool interior = m_base + BM <= p.Cout_g && h0 + p.TH <= p.Ho && w0 + p.TW <= p.Wo;
    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * (TN * 32) + j * 32 + l31;
            const int ho = h0 + (nl >> p.TWlog2), wo = w0 + (nl & (p.TW - 1));
            const bool col_ok = FULL || (ho < p.Ho && wo < p.Wo);
            const long y_col = y_base + (long)ho * p.y_sh + wo, r_col = r_base + (long)ho * p.r_sh + wo;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m0 = m_base + wm * (TM * 32) + i * 32 + 4 * half;
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
                    if (GEN) {
                        p.y[y_base + out_index(p, g * p.Cout_g + m, ho, wo, p.y_sc, p.y_sh)] = combine(p, acc[i][j][r], rv[r], yv[r]);
                    } else {
                        float v = acc[i][j][r];
                        if (p.res_first) v += rv[r];
                        v = apply_act(v, p.act, p.act_slope);
                        if (!p.res_first) v += rv[r];
                        p.y[y_col + (long)(g * p.Cout_g + m) * p.y_sc] = v * p.out_scale + yv[r];
                    }
                }
            }
        }
    };
    if (interior) epilogue(std::true_type{}); else epilogue(std::false_type{});
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
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m_base + i * 16 + q * 4 + r;
            float b = 0.f;
            if (p.bias) {
                const float t = p.bias[g * p.Cout_g + (m < p.Cout_g ? m : 0)];
                b = m < p.Cout_g ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j][r] = b;
        }
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
    const long y_base = (long)n * p.y_sn, r_base = (long)n * p.r_sn;
    const bool interior = m_base + BM <= p.Cout_g && h0 + p.TH <= p.Ho && w0 + p.TW <= p.Wo;
    auto epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * 64 + j * 16 + r16;
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
                if (GEN) {
                    p.y[y_base + out_index(p, g * p.Cout_g + m, ho, wo, p.y_sc, p.y_sh)] = combine(p, acc[e >> 2][j][e & 3], rv[e], yv[e]);
                } else {
                    float v = acc[e >> 2][j][e & 3];
                    if (p.res_first) v += rv[e];
                    v = apply_act(v, p.act, p.act_slope);
                    if (!p.res_first) v += rv[e];
                    p.y[y_col + (long)(g * p.Cout_g + m) * p.y_sc] = v * p.out_scale + yv[e];
                }
            }
        }
    };
    if (interior) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

// ---- pointwise streaming kernel -------------------------------------------------------------------------------------------
// 1x1 layers with at most 8 input or 8 output channels (MDX-Net's 4 -> 48 stem and 48 -> 4 head: 2.4 GB of activations each,
// 2 x 4 x 48 flop per position) are pure HBM streams; as MFMA tiles they ran at 0.45 TB/s.  One thread = 4 consecutive positions
// (float4); the weights are wave-uniform scalar loads from the packed [Cin_pad][Mpad] image.
template <bool FEW_IN>
__global__ void __launch_bounds__(256) conv_pointwise_kernel(ConvArgs p) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;   // float4 index within one image (H * W / 4 of them)
    const long per_img = (long)p.Ho * (p.Wo >> 2);
    if (q >= per_img) return;
    const int n = blockIdx.y;
    const int h = (int)(q / (p.Wo >> 2));
    const int w4 = (int)(q - (long)h * (p.Wo >> 2)) * 4;
    const float* xp = p.x + (long)n * p.x_sn + (long)h * p.x_sh + w4;
    float* yp = p.y + (long)n * p.y_sn + (long)h * p.y_sh + w4;
    const float* rp = p.res ? p.res + (long)n * p.r_sn + (long)h * p.r_sh + w4 : nullptr;
    auto finish = [&](float4 a, int co) __attribute__((always_inline)) {
        if (rp) {
            const float4 r = *reinterpret_cast<const float4*>(rp + (long)co * p.r_sc);
            if (p.res_first) { a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
            a.x = apply_act(a.x, p.act, p.act_slope); a.y = apply_act(a.y, p.act, p.act_slope);
            a.z = apply_act(a.z, p.act, p.act_slope); a.w = apply_act(a.w, p.act, p.act_slope);
            if (!p.res_first) { a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
        } else {
            a.x = apply_act(a.x, p.act, p.act_slope); a.y = apply_act(a.y, p.act, p.act_slope);
            a.z = apply_act(a.z, p.act, p.act_slope); a.w = apply_act(a.w, p.act, p.act_slope);
        }
        a.x *= p.out_scale; a.y *= p.out_scale; a.z *= p.out_scale; a.w *= p.out_scale;
        *reinterpret_cast<float4*>(yp + (long)co * p.y_sc) = a;
    };
    if (FEW_IN) {
        float4 xv[8];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
            xv[ci] = ci < p.Cin_g ? *reinterpret_cast<const float4*>(xp + (long)ci * p.x_sc) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int co = 0; co < p.Cout_g; ++co) {
            const float b = p.bias ? p.bias[co] : 0.f;
            float4 a = make_float4(b, b, b, b);
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (ci < p.Cin_g) {
                    const float wv = p.w[ci * p.Mpad + co];
                    a.x += wv * xv[ci].x; a.y += wv * xv[ci].y; a.z += wv * xv[ci].z; a.w += wv * xv[ci].w;
                }
            }
            finish(a, co);
        }
    } else {
        float4 a[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            const float b = (p.bias && co < p.Cout_g) ? p.bias[co] : 0.f;
            a[co] = make_float4(b, b, b, b);
        }
        for (int ci = 0; ci < p.Cin_g; ++ci) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + (long)ci * p.x_sc);
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                if (co < p.Cout_g) {
                    const float wv = p.w[ci * p.Mpad + co];
                    a[co].x += wv * xv.x; a[co].y += wv * xv.y; a[co].z += wv * xv.z; a[co].w += wv * xv.w;
                }
            }
        }
#pragma unroll
        for (int co = 0; co < 8; ++co)
            if (co < p.Cout_g) finish(a[co], co);
    }
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static unsigned div_mul(int d) { return (unsigned)((0x100000000ULL + (unsigned long long)d - 1) / (unsigned long long)d); }

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
static int choose_tile_width(const ConvArgs& p, int BN) {
    int TW = 1 << ilog2(p.Wo);
    if (TW > BN) TW = BN;
    if (p.Ho == 1) return BN;
    static const bool square = getenv("AICG_CONV_SQUARE") ? atoi(getenv("AICG_CONV_SQUARE")) != 0 : true;
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
    // hoisted producer offsets are 32-bit: one channel chunk of the input / one stage of packed weights must span < 2^31 elements
    const bool off_ok = (long)p.BKC * p.x_sc + (long)p.H * p.x_sh < (1L << 31) && (long)p.taps * p.Cin_pad * p.Mpad < (1L << 31);
    if (lds > 160 * 1024 || p.xs_total > 12 * 256 || !off_ok || (long)p.xs_total * p.CHS >= (1L << 32)) return 1;
    const long gx = (long)p.N * p.tiles_h * p.tiles_w;
    if (gx > 2147483647L) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    dim3 grid((unsigned)gx, (unsigned)idiv_up(p.Cout_g, BM), (unsigned)p.groups);
    dim3 block(64 * (WM * WN + 4));
    const bool gen = p.shuffle || p.res_mul;
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
    static const int ws = getenv("AICG_CONV_WS") ? atoi(getenv("AICG_CONV_WS")) : 1;
    const bool off_ok = (long)p.BKC * p.x_sc + (long)p.H * p.x_sh < (1L << 31) && (long)p.taps * p.Cin_pad * p.Mpad < (1L << 31);
    if (ws && off_ok) {  // wave-specialised form (KSTAGE rows per stage, double-buffered)
        const int xrw = xr <= 8 ? 8 : 12;
        const size_t ldsw = (size_t)(2 * xrw * 256 + 2 * WsGeom<BM, KSTAGE>::WS_ELEMS) * sizeof(float);
        const bool gen = p.shuffle || p.res_mul;
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

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_conv_bkc(int taps) {
    // granularity to which the packed weights pad the input-channel axis (the kernel picks its K chunk <= this)
    (void)taps;
    return 32;
}

extern "C" int aicg_conv_desc_size(void) { return (int)sizeof(aicg_conv_desc); }

extern "C" int aicg_conv_forward(const aicg_conv_desc* d, const float* x, const float* w_packed, const float* bias,
                                 const float* res, float* y, void* stream) {
    if (!d || !x || !w_packed || !y) return fail(AICG_E_ARG, "aicg_conv_forward: null pointer");
    if (d->groups < 1 || d->Cin % d->groups || d->Cout % d->groups)
        return fail(AICG_E_SHAPE, "aicg_conv_forward: channels not divisible by groups");
    if (d->KH < 1 || d->KW < 1 || d->stride_h < 1 || d->stride_w < 1 || d->dil_h < 1 || d->dil_w < 1)
        return fail(AICG_E_SHAPE, "aicg_conv_forward: bad kernel geometry");
    const int pad_h_end = d->pad_h_end < 0 ? d->pad_h : d->pad_h_end, pad_w_end = d->pad_w_end < 0 ? d->pad_w : d->pad_w_end;
    int Ho = (d->H + d->pad_h + pad_h_end - d->dil_h * (d->KH - 1) - 1) / d->stride_h + 1;
    int Wo = (d->W + d->pad_w + pad_w_end - d->dil_w * (d->KW - 1) - 1) / d->stride_w + 1;
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
    p.shuffle = d->shuffle; p.res_mul = d->res_mul;
    if (p.shuffle != 0 && (p.shuffle != 2 || d->Cout % 4 || d->groups != 1))
        return fail(AICG_E_ARG, "aicg_conv_forward: shuffle must be 0 or 2 (Cout %% 4 == 0, groups == 1)");
    if (p.res_mul && (!res || d->res_before_act)) return fail(AICG_E_ARG, "aicg_conv_forward: res_mul needs res and res_before_act == 0");
    p.taps = p.KH * p.KW;
    p.Mpad = idiv_up(p.Cout_g, 32) * 32;
    p.Cin_pad = idiv_up(p.Cin_g, 32) * 32;
    p.w_group_stride = (long)p.taps * p.Cin_pad * p.Mpad;

    // pointwise streaming form: 1x1, unit stride, no padding, <= 8 channels on one side, float4-aligned rows
    {
        static const bool pointwise = getenv("AICG_CONV_POINTWISE") ? atoi(getenv("AICG_CONV_POINTWISE")) != 0 : true;
        const bool few_in = p.Cin_g <= 8, few_out = p.Cout_g <= 8;
        auto al4 = [](long v) { return (v & 3) == 0; };
        const bool aligned = (Wo & 3) == 0 && al4(p.x_sn) && al4(p.x_sc) && al4(p.x_sh) && al4(p.y_sn) && al4(p.y_sc) && al4(p.y_sh) &&
                             (!res || (al4(p.r_sn) && al4(p.r_sc) && al4(p.r_sh) && ((uintptr_t)res & 15) == 0)) &&
                             ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0;
        if (pointwise && p.taps == 1 && p.groups == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && pad_h_end == 0 &&
            pad_w_end == 0 && (few_in || few_out) && p.Cin_g <= 512 && p.Cout_g <= 512 && p.pre_act == AICG_ACT_NONE &&
            !p.accumulate && aligned && Ho == p.H && Wo == p.W && !p.shuffle && !p.res_mul) {
            const long per_img = (long)Ho * (Wo >> 2);
            dim3 grid((unsigned)ldiv_up(per_img, 256), (unsigned)p.N);
            if (few_in) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_pointwise_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, p);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_pointwise_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, p);
            return check_launch("conv_pointwise_kernel");
        }
    }

    // tile shape: the BM in {160, 128, 96, 64, 32} with the least padded M (larger BM on ties: one input patch
    // staging feeds more MFMAs), then enough workgroups to fill 256 CUs
    const int M = p.Cout_g;
    int BM = 32;
    {
        long best = 1L << 40;
        const int cands[5] = {160, 128, 96, 64, 32};
        for (int i = 0; i < 5; ++i) {
            const long padded = (long)idiv_up(M, cands[i]) * cands[i];
            if (padded < best) { best = padded; BM = cands[i]; }
        }
    }
    const long npos = (long)p.N * Ho * Wo;
    hipStream_t st = (hipStream_t)stream;
    static const int ablate = getenv("AICG_CONV_ABLATE") ? atoi(getenv("AICG_CONV_ABLATE")) : 0;
    p.dbg = ablate;
    // narrow layers (16 / 48 output channels, at least 3 input channels, enough positions): 16x16x4 MFMA tiles
    static const bool use16 = getenv("AICG_CONV_M16") ? atoi(getenv("AICG_CONV_M16")) != 0 : true;
    if (use16 && p.Cin_g >= 3 && npos >= 256L * 256) {
        int rc = 1;
        if (M > 32 && M <= 48) rc = launch_conv16<48>(p, st);
        else if (M <= 16) rc = launch_conv16<16>(p, st);
        if (rc <= 0) return rc;   // launched (0) or failed with an error code (< 0); 1 = not applicable
    }
    // A launch should give each of the 256 CUs at least ~2 workgroups: shrink the tile for small problems
    // (HuBERT / enc_p GEMMs over a few thousand frames), M first (keeps the wide, coalesced N tile), then N.
    auto blocks = [&](int bm, int bn) { return (long)idiv_up(M, bm) * p.groups * ldiv_up(npos, bn); };
    const long want = 512;
    static const int ws = getenv("AICG_CONV_WS") ? atoi(getenv("AICG_CONV_WS")) : 1;
    if (ws) {
        int rc = 1;
        if (BM == 160 && blocks(160, 128) >= want) rc = launch_conv_ws<160, 128, 1, 4, 32>(p, st);
        else if (BM == 128 && blocks(128, 128) >= want) {
            // >= 2 M tiles of 128: four-consumer 64-row tiles (two workgroups per CU overlap their prologue / epilogue) measured
            // 7 % faster than the one-per-CU 128 x 128 shape on the 256-channel vocoder stage; a single 128-row tile keeps the latter
            static const int bm128e = getenv("AICG_CONV_BM128") ? atoi(getenv("AICG_CONV_BM128")) : 0;
            const bool use64 = bm128e ? bm128e == 64 : M >= 256;
            rc = use64 ? launch_conv_ws<64, 128, 2, 2, 64>(p, st) : launch_conv_ws<128, 128, 2, 4, 64>(p, st);
        }
        else if (BM == 96 && blocks(96, 128) >= want) rc = launch_conv_ws<96, 128, 1, 4, 64>(p, st);
        else if (M > 32 && blocks(64, 128) >= want) rc = launch_conv_ws<64, 128, 2, 2, 64>(p, st);
        if (rc <= 0) return rc;
        // small problems (few output positions: HuBERT / enc_p / flow GEMMs) and 32-channel layers: the same kernel on smaller tiles
        static const int smallws = getenv("AICG_CONV_SMALLWS") ? atoi(getenv("AICG_CONV_SMALLWS")) : 1;
        if (smallws) {
            if (M > 32) {
                if (!(blocks(64, 128) >= want) && (blocks(64, 64) >= want || M > 64)) rc = launch_conv_ws<64, 64, 2, 2, 64>(p, st);
            } else if (blocks(32, 256) >= want) rc = launch_conv_ws<32, 256, 1, 4, 64>(p, st);
            else rc = launch_conv_ws<32, 128, 1, 4, 64>(p, st);
            if (rc <= 0) return rc;
        }
    }
    // single-role kernels: layers the wave-specialised form could not stage.  A patch that is too large even here (wide
    // strided / dilated 2-D windows) is retried on the narrowest tile (64 positions) before giving up.
    static const bool eight = getenv("AICG_CONV_8WAVE") ? atoi(getenv("AICG_CONV_8WAVE")) != 0 : true;
    auto single_role = [&]() -> int {
        if (BM == 160 && blocks(160, 128) >= want) return launch_conv<160, 128, 1, 4>(p, st);
        if (BM == 128 && blocks(128, 128) >= want) return eight ? launch_conv<128, 128, 2, 4>(p, st) : launch_conv<128, 128, 2, 2>(p, st);
        if (BM == 96 && blocks(96, 128) >= want) return launch_conv<96, 128, 1, 4>(p, st);
        if (M > 32) {
            if (blocks(64, 128) >= want) return launch_conv<64, 128, 2, 2>(p, st);
            if (blocks(64, 64) >= want || M > 64) return launch_conv<64, 64, 2, 2>(p, st);
        }
        if (blocks(32, 256) >= want) return launch_conv<32, 256, 1, 4>(p, st);
        return launch_conv<32, 128, 1, 4>(p, st);
    };
    int rc = single_role();
    if (rc == AICG_E_LDS) rc = launch_conv<64, 64, 2, 2>(p, st);
    return rc;
}
