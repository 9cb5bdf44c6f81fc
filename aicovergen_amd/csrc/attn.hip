// Fused softmax attention on fp32 MFMA for channel-major (C, T) activations; T x T scores never reach HBM.
//   * enc_p MultiHeadAttention with windowed relative-position keys/values (reference
//     src/infer_pack/attentions.py:226-275; closed form: SURVEY appendix B.3) -- 2 heads x 96
//   * HuBERT self-attention (fairseq MultiheadAttention, called at src/vc_infer_pipeline.py:398-406) -- 12 x 64
//
// One wave owns 32 queries.  S^T = K Q^T is computed "swapped" (A = K tile from LDS, B = Q from
// registers) so that, in the 32x32 MFMA accumulator layout, a lane holds 16 keys of ONE query column:
// the row max / row sum are lane-local plus one xor-32 shuffle, and the rescale factor of the running
// output is a per-lane scalar.  O^T += V^T P^T then uses the S^T accumulator registers directly as the B
// operand: MFMA step r pairs key j_r (lanes 0-31) with key j_r + 4 (lanes 32-63), which is exactly the
// pair of rows register r holds in the two half-waves -- no cross-lane movement of P.
#include "common.h"

namespace aicg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
    const float* q;
    const float* k;
    const float* v;
    const float* relk;  // (H, 2w+1, T) precomputed q_i . E^k_m, or null
    float* o;
    float* lse;  // (H, T) log-sum-exp per query, or null
    int T, H, window;
    long ldq, ldk, ldv, ldo;  // row (channel) strides
    float scale;
    int nsplit;     // key range split over gridDim.z (split-K): partial (o, m, l) go to part, merged by attn_combine_kernel
    float* part;    // [nsplit][H][D][T] un-normalised outputs, then [nsplit][H][T] running max, then [nsplit][H][T] running sum
};

static constexpr int KT = 32;       // keys per tile
static constexpr int V_LD4 = 9;     // float4 per V row in LDS (36 floats: 16-byte aligned rows, b128 reads of 16 consecutive rows hit 16
                                    // distinct bank quads: 9 d mod 16 is a permutation)
// Fragment layouts (r3; the convolution family's finding applies here too: every hand-over of LDS-loaded registers to the matrix
// pipe costs ~50-64 cycles, so ONE ds_read_b128 per fragment feeds FOUR MFMA k-steps instead of one ds_read_b32 each):
//   K tile   [d / 8][d & 1][key][4]   element e of the quad = channel 8 (d / 8) + 2 e + (d & 1): lane (key, half) reads plane
//            (g, half) and owns the A operands of k-steps 4 g .. 4 g + 3 of S^T = K Q^T (k-step s contracts channels 2 s, 2 s + 1);
//   V tile   [d][key], 36-float rows: lane (d, half) reads keys 8 q + 4 half .. + 3 -- exactly the key pairs MFMA steps
//            4 q .. 4 q + 3 of O^T += V^T P^T contract against the S^T accumulator registers 4 q .. 4 q + 3.

// Key tiles are software-pipelined: the K/V tile kt + 1 is loaded into registers while tile kt is consumed from LDS.
// Few (query block, head) pairs exist for long single-head-group sequences (enc_p: 52 x 2), so the key range can be split over
// gridDim.z; each split keeps its own running (max, sum) and attn_combine_kernel merges them (the usual log-sum-exp merge).
// r4 (tools/kbench_attn_ab.py, profiles/r04_kbench_attn_ab.txt; HuBERT 12 x 64 at T = 3300 / enc_p 2 x 96 at T = 6600):
//   OCC   waves per SIMD the register allocation is held to: left alone hipcc took 188 registers at D = 64 and 257 at D = 96 -- two and
//         ONE wave per SIMD; 158 and 218 cost no spill and admit three and two (0.437 -> 0.418 ms / 0.777 -> 0.673 ms; one more wave
//         each spills and loses 20 %).
//   The softmax runs in the base-2 domain: log2 e is folded into the query scale (and the relative-key bias), a score costs one
//   subtraction and one v_exp_f32 instead of expf's range reduction (-> 0.381 / 0.633 ms), and the running output is rescaled only on
//   tiles where some lane's maximum moved (alpha is exactly 1 on every lane otherwise; -> 0.625 ms at D = 96).  Error against a float64
//   softmax unchanged (5.6e-7 / 7.7e-7).  lse and the split-K partial maxima are converted back to the natural domain.
__device__ __forceinline__ float attn_exp2(float x) {
#ifdef AICG_EMULATED
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);   // v_exp_f32: 1 ulp, exp2(-inf) = 0
#endif
}

template <int D, int OCC>
__global__ void __launch_bounds__(256) AICG_WAVES_PER_SIMD(OCC) attn_fwd_kernel(AttnArgs p) {
    constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    constexpr int DT = D / 32;
    constexpr int NL = D * KT / 256;  // K (and V) elements each thread stages per tile
    constexpr int NQ = NL / 4;        // K quads (4 channels of one key) per thread
    static_assert(NL % 4 == 0, "a thread stages whole K quads");
    __shared__ __attribute__((aligned(16))) float4 Ks4[(D / 4) * KT];     // (D / 8) x 2 planes x 32 keys
    __shared__ __attribute__((aligned(16))) float4 Vs4[D * V_LD4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y;
    const int i0 = blockIdx.x * 128 + wave * 32;  // first query of this wave
    const int qi = i0 + l31;
    const float* qh = p.q + (long)h * D * p.ldq;
    const float* kh = p.k + (long)h * D * p.ldk;
    const float* vh = p.v + (long)h * D * p.ldv;

    // Q^T fragments: qreg[s] = q[2s + half][qi]
    float qreg[D / 2];
    const float qscale = p.scale * kLog2e;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) qreg[s] = (qi < p.T) ? qh[(long)(2 * s + half) * p.ldq + qi] * qscale : 0.f;

    f32x16 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = idiv_up(p.T, KT);
    const int per = idiv_up(ntiles, p.nsplit);
    const int kt_begin = blockIdx.z * per, kt_end = imin(ntiles, kt_begin + per);
    const int blk_q0 = blockIdx.x * 128;
    float4 kpre[NQ];
    float vpre[NL];
    auto prefetch = [&](int kt) {
        const int j0 = kt * KT;
#pragma unroll
        for (int e = 0; e < NQ; ++e) {   // K: quad (plane = 2 g + parity, key j): channels 8 g + parity + {0, 2, 4, 6}, coalesced along j
            const int item = tid + e * 256;
            const int plane = item >> 5, j = item & 31;
            const bool ok = (j0 + j) < p.T;
            const float* src = kh + (long)(8 * (plane >> 1) + (plane & 1)) * p.ldk + (ok ? j0 + j : 0);
            const float t0 = src[0], t1 = src[2 * p.ldk], t2 = src[4 * p.ldk], t3 = src[6 * p.ldk];
            kpre[e] = ok ? make_float4(t0, t1, t2, t3) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int e = 0; e < NL; ++e) {
            const int idx = tid + e * 256;
            const int d = idx >> 5, j = idx & 31;
            const bool ok = (j0 + j) < p.T;
            const float tv = vh[(long)d * p.ldv + (ok ? j0 + j : 0)];
            vpre[e] = ok ? tv : 0.f;
        }
    };
    if (kt_begin < kt_end) prefetch(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int j0 = kt * KT;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NQ; ++e) Ks4[tid + e * 256] = kpre[e];          // item index = (plane, key): the layout itself
#pragma unroll
        for (int e = 0; e < NL; ++e) {
            const int idx = tid + e * 256;
            reinterpret_cast<float*>(Vs4)[(idx >> 5) * (4 * V_LD4) + (idx & 31)] = vpre[e];
        }
        __syncthreads();
        if (kt + 1 < kt_end) prefetch(kt + 1);
        // S^T tile: rows = keys, cols = queries
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        {
            const float4* kf = Ks4 + half * KT + l31;
            float4 a = kf[0];
#pragma unroll
            for (int g = 0; g < D / 8; ++g) {
                const float4 an = kf[(g + 1 < D / 8 ? g + 1 : g) * 2 * KT];     // next k-group's quad in flight under these MFMAs
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * g], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * g + 1], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * g + 2], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * g + 3], st, 0, 0, 0);
                a = an;
            }
        }
        // relative-position key bias on the band |j - i| <= window (only near-diagonal tiles)
        if (p.relk && j0 + KT - 1 >= blk_q0 - p.window && j0 <= blk_q0 + 127 + p.window) {
            const float* rk = p.relk + (long)h * (2 * p.window + 1) * p.T;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int dlt = j - qi;
                if (qi < p.T && j < p.T && dlt >= -p.window && dlt <= p.window)
                    st[r] += rk[(long)(dlt + p.window) * p.T + qi] * kLog2e;
            }
        }
        // online softmax over this tile's 32 keys of query qi
        float m_t = -INFINITY;
        if (j0 + KT > p.T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (j >= p.T) st[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) m_t = fmaxf(m_t, st[r]);
        m_t = fmaxf(m_t, __shfl_xor(m_t, 32));
        const float m_new = fmaxf(m_run, m_t);
        const float alpha = attn_exp2(m_run - m_new);
        const bool moved = __ballot(m_new > m_run) != 0ull;   // (alpha == 1 exactly on every lane otherwise)
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = attn_exp2(st[r] - m_new);
            rs += st[r];
        }
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        if (moved) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
        }
        // O^T += V^T P^T : step r contracts keys (j_r, j_r + 4), j_r = (r & 3) + 8 (r >> 2); the quad q = r >> 2 of a V row is one float4
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) a[dt] = Vs4[(dt * 32 + l31) * V_LD4 + 2 * q + half];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt].x, st[4 * q], acc[dt], 0, 0, 0);
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt].y, st[4 * q + 1], acc[dt], 0, 0, 0);
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt].z, st[4 * q + 2], acc[dt], 0, 0, 0);
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt].w, st[4 * q + 3], acc[dt], 0, 0, 0);
            }
        }
    }
    if (qi >= p.T) return;
    if (p.nsplit > 1) {
        const long sh = (long)blockIdx.z * p.H + h;
        float* po = p.part + sh * D * p.T;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                po[(long)d * p.T + qi] = acc[dt][r];
            }
        if (half == 0) {
            float* pm = p.part + (long)p.nsplit * p.H * D * p.T;
            pm[sh * p.T + qi] = m_run * kLn2;   // the merge pass works in the natural domain
            pm[((long)p.nsplit * p.H + sh) * p.T + qi] = l_run;
        }
        return;
    }
    const float inv = 1.f / l_run;
    float* oh = p.o + (long)h * D * p.ldo;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            oh[(long)d * p.ldo + qi] = acc[dt][r] * inv;
        }
    if (p.lse && half == 0) p.lse[(long)h * p.T + qi] = m_run * kLn2 + logf(l_run);
}

// merge of the split-K partials: o = sum_s acc_s e^{m_s - m} / sum_s l_s e^{m_s - m}; grid (T/256, H, D/16)
__global__ void __launch_bounds__(256) attn_combine_kernel(AttnArgs p, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y;
    if (i >= p.T) return;
    const float* pm = p.part + (long)p.nsplit * p.H * D * p.T;
    const float* pl = pm + (long)p.nsplit * p.H * p.T;
    float m = -INFINITY;
    for (int s = 0; s < p.nsplit; ++s) m = fmaxf(m, pm[((long)s * p.H + h) * p.T + i]);
    float w[16];
    float l = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        w[s] = 0.f;
        if (s < p.nsplit) {
            const long sh = (long)s * p.H + h;
            w[s] = expf(pm[sh * p.T + i] - m);
            l += pl[sh * p.T + i] * w[s];
        }
    }
    const float inv = 1.f / l;
    float* oh = p.o + (long)h * D * p.ldo;
    const int d0 = blockIdx.z * 16;
#pragma unroll 4
    for (int d = d0; d < d0 + 16; ++d) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s < p.nsplit) a += p.part[(((long)s * p.H + h) * D + d) * p.T + i] * w[s];
        oh[(long)d * p.ldo + i] = a * inv;
    }
    if (p.lse && blockIdx.z == 0) p.lse[(long)h * p.T + i] = m + logf(l);
}

// Relative-position VALUE term: o_i += sum_{|j-i|<=w} P_ij E^v_{j-i+w}, with P rebuilt from the saved
// log-sum-exp (attentions.py:264-271).  A workgroup owns 64 queries of one head: the key columns i0 - w .. i0 + 63 + w of all D channels
// are staged in LDS once (the band of a query is 2w + 1 of them), four waves each contract a quarter of the channels into the
// 2w + 1 band scores of their query (partial sums met in LDS), the probabilities are formed once per (query, offset), and the four
// waves each add a quarter of the D output channels.  (The first form -- one thread per query, 64-thread workgroups, every key read
// from global memory per channel -- took 208 us for T = 6 600, two heads: 2 016 dependent loads per thread on 208 waves.)
template <int D, int NW>
__global__ void __launch_bounds__(256) attn_relv_kernel(AttnArgs p, const float* __restrict__ relv_emb) {
    constexpr int KC = 64 + NW - 1;             // staged key columns
    constexpr int DQ = D / 4;                   // channels per wave
    __shared__ float Ev[NW * D];
    __shared__ float Ks[D * KC];
    __shared__ float Ss[4 * NW * 64];           // [channel quarter][offset][query]; quarter 0 becomes the probabilities
    const int tid = threadIdx.x;
    const int qi = tid & 63, dg = tid >> 6;
    const int i0 = blockIdx.x * 64, i = i0 + qi;
    const int h = blockIdx.y;
    const int w = p.window;
    const float* qh = p.q + (long)h * D * p.ldq;
    const float* kh = p.k + (long)h * D * p.ldk;
    for (int idx = tid; idx < NW * D; idx += 256) Ev[idx] = relv_emb[idx];
    for (int idx = tid; idx < D * KC; idx += 256) {
        const int d = idx / KC, c = idx - d * KC;
        const int j = i0 - w + c;
        Ks[idx] = (j >= 0 && j < p.T) ? kh[(long)d * p.ldk + j] : 0.f;
    }
    __syncthreads();
    float s[NW];
#pragma unroll
    for (int m = 0; m < NW; ++m) s[m] = 0.f;
#pragma unroll 2
    for (int dd = 0; dd < DQ; ++dd) {
        const int d = dg * DQ + dd;
        const float qd = i < p.T ? qh[(long)d * p.ldq + i] * p.scale : 0.f;
        const float* kr = Ks + d * KC + qi;
#pragma unroll
        for (int m = 0; m < NW; ++m) s[m] += qd * kr[m];
    }
#pragma unroll
    for (int m = 0; m < NW; ++m) Ss[(dg * NW + m) * 64 + qi] = s[m];
    __syncthreads();
    const float* rk = p.relk + (long)h * NW * p.T;
    for (int idx = tid; idx < NW * 64; idx += 256) {
        const int m = idx >> 6, q = idx & 63;
        const int iq = i0 + q, j = iq + m - w;
        const float tot = (Ss[idx] + Ss[NW * 64 + idx]) + (Ss[2 * NW * 64 + idx] + Ss[3 * NW * 64 + idx]);
        Ss[idx] = (iq < p.T && j >= 0 && j < p.T) ? expf(tot + rk[(long)m * p.T + iq] - p.lse[(long)h * p.T + iq]) : 0.f;
    }
    __syncthreads();
    if (i >= p.T) return;
#pragma unroll
    for (int m = 0; m < NW; ++m) s[m] = Ss[m * 64 + qi];
    float* oh = p.o + (long)h * D * p.ldo;
#pragma unroll 2
    for (int dd = 0; dd < DQ; ++dd) {
        const int d = dg * DQ + dd;
        float a = 0.f;
#pragma unroll
        for (int m = 0; m < NW; ++m) a += s[m] * Ev[m * D + d];
        oh[(long)d * p.ldo + i] += a;
    }
}

}  // namespace aicg

using namespace aicg;

static int attention_launch(AttnArgs& p, int D, hipStream_t stream) {
    dim3 grid((unsigned)idiv_up(p.T, 128), (unsigned)p.H, (unsigned)p.nsplit);
    if (D == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_fwd_kernel<64, 3>), grid, dim3(256), 0, stream, p);
    else if (D == 96) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_fwd_kernel<96, 2>), grid, dim3(256), 0, stream, p);
    else if (D == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_fwd_kernel<32, 4>), grid, dim3(256), 0, stream, p);
    else if (D == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_fwd_kernel<128, 1>), grid, dim3(256), 0, stream, p);
    else return fail(AICG_E_SHAPE, "aicg_attention: head dim %d not in {32,64,96,128}", D);
    int rc = check_launch("attn_fwd_kernel");
    if (rc != AICG_OK || p.nsplit == 1) return rc;
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)idiv_up(p.T, 256), (unsigned)p.H, (unsigned)(D / 16)), dim3(256), 0, stream, p, D);
    return check_launch("attn_combine_kernel");
}

extern "C" int aicg_attention(const float* q, const float* k, const float* v, const float* relk, float* o, float* lse, int T,
                              int H, int D, int window, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                              void* stream) {
    if (!q || !k || !v || !o) return fail(AICG_E_ARG, "aicg_attention: null pointer");
    if (T < 1 || H < 1) return fail(AICG_E_SHAPE, "aicg_attention: bad shape");
    AttnArgs p{q, k, v, relk, o, lse, T, H, window, (long)ldq, (long)ldk, (long)ldv, (long)ldo, scale, 1, nullptr};
    return attention_launch(p, D, (hipStream_t)stream);
}

extern "C" int aicg_attention_split(const float* q, const float* k, const float* v, const float* relk, float* o, float* lse,
                                    int T, int H, int D, int window, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                    float scale, int n_splits, float* scratch, void* stream) {
    if (!q || !k || !v || !o) return fail(AICG_E_ARG, "aicg_attention_split: null pointer");
    if (T < 1 || H < 1) return fail(AICG_E_SHAPE, "aicg_attention_split: bad shape");
    if (n_splits < 1 || n_splits > 16 || n_splits > idiv_up(T, KT))
        return fail(AICG_E_ARG, "aicg_attention_split: n_splits %d not in [1, min(16, key tiles)]", n_splits);
    if (n_splits > 1 && !scratch) return fail(AICG_E_ARG, "aicg_attention_split: scratch is required for n_splits > 1");
    AttnArgs p{q, k, v, relk, o, lse, T, H, window, (long)ldq, (long)ldk, (long)ldv, (long)ldo, scale, n_splits, scratch};
    return attention_launch(p, D, (hipStream_t)stream);
}

extern "C" int aicg_attention_relv(const float* q, const float* k, const float* relk, const float* relv_emb,
                                   const float* lse, float* o, int T, int H, int D, int window, int64_t ldq, int64_t ldk,
                                   int64_t ldo, float scale, void* stream) {
    if (!q || !k || !relk || !relv_emb || !lse || !o) return fail(AICG_E_ARG, "aicg_attention_relv: null pointer");
    AttnArgs p{q, k, nullptr, relk, o, const_cast<float*>(lse), T, H, window, (long)ldq, (long)ldk, 0, (long)ldo, scale, 1, nullptr};
    dim3 grid((unsigned)idiv_up(T, 64), (unsigned)H);
    if (D == 96 && window == 10)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_relv_kernel<96, 21>), grid, dim3(256), 0, (hipStream_t)stream, p, relv_emb);
    else if (D == 64 && window == 10)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_relv_kernel<64, 21>), grid, dim3(256), 0, (hipStream_t)stream, p, relv_emb);
    else if (D == 32 && window == 10)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_relv_kernel<32, 21>), grid, dim3(256), 0, (hipStream_t)stream, p, relv_emb);
    else if (D == 32 && window == 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(attn_relv_kernel<32, 9>), grid, dim3(256), 0, (hipStream_t)stream, p, relv_emb);
    else return fail(AICG_E_SHAPE, "aicg_attention_relv: (D=%d, window=%d) not instantiated", D, window);
    return check_launch("attn_relv_kernel");
}
