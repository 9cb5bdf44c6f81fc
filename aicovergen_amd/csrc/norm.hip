// Normalisation kernels (HBM-bound, two-pass statistics in fp32 like torch):
//   * layernorm_ct : LayerNorm over channels of a channel-major (C, T) map, optional fused residual add
//                    (reference modules.LayerNorm, src/infer_pack/modules.py:29-32, called from
//                     attentions.Encoder.forward :61-73 as norm(x + y); HuBERT encoder LayerNorms)
//   * rownorm_act  : per-row mean/variance over time + affine + activation: HuBERT feature-extractor
//                    GroupNorm(512 groups == 512 channels) + GELU (fairseq ConvFeatureExtractionModel layer 0)
#include "common.h"

#include <cstdint>

namespace aicg {

// block = 64 time columns x 4 channel groups; every wave reads 256-byte rows
__global__ void __launch_bounds__(256) layernorm_ct_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ out, int C, long T, float eps,
                                                           long x_sn, long r_sn, long o_sn) {
    __shared__ float red[4][64];
    __shared__ float stat[2][64];
    const int tx = threadIdx.x & 63, cy = threadIdx.x >> 6;
    const long t = (long)blockIdx.x * 64 + tx;
    const int n = blockIdx.y;
    const bool ok = t < T;
    const float* xn = x + (long)n * x_sn;
    const float* rn = res ? res + (long)n * r_sn : nullptr;
    float s = 0.f;
    if (ok)
        for (int c = cy; c < C; c += 4) s += xn[(long)c * T + t] + (rn ? rn[(long)c * T + t] : 0.f);
    red[cy][tx] = s;
    __syncthreads();
    if (cy == 0) stat[0][tx] = (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / (float)C;
    __syncthreads();
    const float mean = stat[0][tx];
    float v = 0.f;
    if (ok)
        for (int c = cy; c < C; c += 4) {
            const float d = xn[(long)c * T + t] + (rn ? rn[(long)c * T + t] : 0.f) - mean;
            v += d * d;
        }
    __syncthreads();
    red[cy][tx] = v;
    __syncthreads();
    if (cy == 0) stat[1][tx] = rsqrtf((red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / (float)C + eps);
    __syncthreads();
    const float rstd = stat[1][tx];
    if (ok) {
        float* on = out + (long)n * o_sn;
        for (int c = cy; c < C; c += 4) {
            const float val = xn[(long)c * T + t] + (rn ? rn[(long)c * T + t] : 0.f);
            on[(long)c * T + t] = (val - mean) * rstd * gamma[c] + beta[c];
        }
    }
}

// Register-resident form for C <= G * NV: block = CB time columns x G = 1024 / CB channel groups, every element is read once, kept
// in registers for the two-pass mean / variance and written once (the strided form above re-reads it three times from 52
// workgroups for a HuBERT layer: 183 us for 10 MB).  CB = 32 reads 128-byte row segments; the maps of one chunk are short
// (T = 3300 frames -> 104 workgroups on 256 CUs), so launches that would leave CUs idle take CB = 8 (32-byte segments of rows
// the neighbouring workgroups fetch at the same time: the sectors meet in L2) and four times the workgroups.
template <int NV, int CB>
__global__ void __launch_bounds__(1024) layernorm_ct_reg_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ out, int C, long T, float eps, long x_sn,
                                                                long r_sn, long o_sn) {
    constexpr int G = 1024 / CB;
    __shared__ float red[G][CB + 1];
    __shared__ float stat[CB];
    const int tx = threadIdx.x % CB, gy = threadIdx.x / CB;
    const long t = (long)blockIdx.x * CB + tx;
    const int n = blockIdx.y;
    const bool ok = t < T;
    const long tt = ok ? t : 0;
    const float* xn = x + (long)n * x_sn;
    const float* rn = res ? res + (long)n * r_sn : nullptr;
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int c = gy + G * e;
        const bool in = c < C;
        const long off = (long)(in ? c : 0) * T + tt;
        float a = xn[off];
        if (rn) a += rn[off];
        v[e] = in ? a : 0.f;
        s += v[e];
    }
    red[gy][tx] = s;
    __syncthreads();
    if (gy == 0) {
        float m = 0.f;
#pragma unroll 8
        for (int k = 0; k < G; ++k) m += red[k][tx];
        stat[tx] = m / (float)C;
    }
    __syncthreads();
    const float mean = stat[tx];
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float d = (gy + G * e < C) ? v[e] - mean : 0.f;
        q += d * d;
    }
    __syncthreads();
    red[gy][tx] = q;
    __syncthreads();
    if (gy == 0) {
        float m = 0.f;
#pragma unroll 8
        for (int k = 0; k < G; ++k) m += red[k][tx];
        stat[tx] = rsqrtf(m / (float)C + eps);
    }
    __syncthreads();
    const float rstd = stat[tx];
    if (!ok) return;
    float* on = out + (long)n * o_sn;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int c = gy + G * e;
        if (c < C) on[(long)c * T + t] = (v[e] - mean) * rstd * gamma[c] + beta[c];
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// one workgroup per row (channel): mean, variance (two passes), then normalise + act
__global__ void __launch_bounds__(256) rownorm_act_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out, long T,
                                                          long ld, float eps, int act) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const float* xr = x + row * ld;
    float s = 0.f;
    for (long t = threadIdx.x; t < T; t += 256) s += xr[t];
    const float mean = block_sum_256(s, sh) / (float)T;
    float v = 0.f;
    for (long t = threadIdx.x; t < T; t += 256) {
        const float d = xr[t] - mean;
        v += d * d;
    }
    const float var = block_sum_256(v, sh) / (float)T;
    const float rstd = rsqrtf(var + eps);
    const float g = gamma ? gamma[row] : 1.f, b = beta ? beta[row] : 0.f;
    float* orow = out + row * ld;
    for (long t = threadIdx.x; t < T; t += 256) orow[t] = apply_act((xr[t] - mean) * rstd * g + b, act, 0.f);
}

// ---- rownorm, split form for long rows (HuBERT layer-0 GroupNorm: 512 rows x 211 231 frames per 66 s chunk) -----------------------
// The one-workgroup-per-row kernel above streams a row three times with 4-byte loads from 512 workgroups.  Here
//   (1) rownorm_stats_kernel: grid (segments, rows); a thread keeps up to 16 float4 of its segment in REGISTERS, takes their exact
//       two-pass mean / M2, and the partial (n, mean, M2) triples are merged pairwise (Chan et al.) over the lanes, the waves and --
//       in (2) -- the segments, always in the same order: one read of x, deterministic, as accurate as the two-pass form;
//   (2) rownorm_apply_kernel: merges the row's partials, then normalise + affine + activation as one float4 stream.
// Rows start at arbitrary 4-byte offsets (T is odd): a row is cut into <= 3 head scalars, an aligned float4 body and <= 3 tail
// scalars; x and out must share their alignment (checked by the host wrapper).  `ld` = row stride of x AND out (T for contiguous maps;
// a multiple of 4 when the caller keeps rows 16-byte aligned for the DMA-staged convolution behind, conv_g1s.h: then every head is 0).
struct Moments { float n, mean, m2; };

__device__ __forceinline__ Moments merge(Moments a, Moments b) {
    const float n = a.n + b.n;
    if (n == 0.f) return a;
    const float d = b.mean - a.mean, fb = b.n / n;
    return Moments{n, a.mean + d * fb, a.m2 + b.m2 + d * d * a.n * fb};
}

struct RowSplit { int head; long nb4; int tail; long q4; };   // q4 = float4 per segment
__device__ __forceinline__ RowSplit row_split(long row, long T, long ld, int nseg) {
    RowSplit r;
    r.head = (int)((4 - ((row * ld) & 3)) & 3);
    if (r.head > T) r.head = (int)T;
    r.nb4 = (T - r.head) >> 2;
    r.tail = (int)((T - r.head) & 3);
    r.q4 = (r.nb4 + nseg - 1) / nseg;
    return r;
}

constexpr int kRnV = 16;   // float4 per thread and segment

__global__ void __launch_bounds__(256) rownorm_stats_kernel(const float* __restrict__ x, float* __restrict__ part, long T, long ld, int nseg) {
    __shared__ Moments sh[4];
    const long row = blockIdx.y;
    const int seg = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + row * ld;
    const RowSplit rs = row_split(row, T, ld, nseg);
    const long b0 = (long)seg * rs.q4, b1 = lmin(b0 + rs.q4, rs.nb4);
    const float4* body = reinterpret_cast<const float4*>(xr + rs.head);
    float4 v[kRnV];
    float s = 0.f, cnt = 0.f;
#pragma unroll
    for (int e = 0; e < kRnV; ++e) {
        const long i = b0 + tid + 256L * e;
        const bool in = i < b1;
        v[e] = in ? body[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
        cnt += in ? 4.f : 0.f;
    }
    // the row's head / tail scalars ride with the first / last segment (one extra value on a few lanes)
    float extra = 0.f;
    bool has_extra = false;
    if (seg == 0 && tid < rs.head) { extra = xr[tid]; has_extra = true; }
    if (seg == nseg - 1 && tid >= 64 && tid - 64 < rs.tail) { extra = xr[rs.head + 4 * rs.nb4 + (tid - 64)]; has_extra = true; }
    if (has_extra) { s += extra; cnt += 1.f; }
    Moments m{cnt, cnt > 0.f ? s / cnt : 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < kRnV; ++e) {
        if (b0 + tid + 256L * e < b1) {
            const float d0 = v[e].x - m.mean, d1 = v[e].y - m.mean, d2 = v[e].z - m.mean, d3 = v[e].w - m.mean;
            m.m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    if (has_extra) { const float d = extra - m.mean; m.m2 += d * d; }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Moments b{__shfl_xor(m.n, o, 64), __shfl_xor(m.mean, o, 64), __shfl_xor(m.m2, o, 64)};
        // the lower lane of a pair keeps the (a, b) order so that both lanes compute the same bits
        m = (tid & o) ? merge(b, m) : merge(m, b);
    }
    if ((tid & 63) == 0) sh[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        const Moments r = merge(merge(sh[0], sh[1]), merge(sh[2], sh[3]));
        float* p = part + (row * nseg + seg) * 3;
        p[0] = r.n; p[1] = r.mean; p[2] = r.m2;
    }
}

template <int ACT>
__device__ __forceinline__ float rn_act(float v, int act) {
    if (ACT == AICG_ACT_NONE) return v;
    if (ACT == AICG_ACT_GELU) return gelu_erf(v);
    return apply_act(v, act, 0.f);
}

template <int ACT>
__global__ void __launch_bounds__(256) rownorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ out, long T, long ld, int nseg, float eps, int act) {
    const long row = blockIdx.y;
    const int seg = blockIdx.x, tid = threadIdx.x;
    Moments m{0.f, 0.f, 0.f};
    for (int q = 0; q < nseg; ++q) {
        const float* p = part + (row * nseg + q) * 3;
        m = merge(m, Moments{p[0], p[1], p[2]});
    }
    const float rstd = rsqrtf(m.m2 / (float)T + eps);
    const float g = gamma ? gamma[row] : 1.f, b = beta ? beta[row] : 0.f;
    // (x - mean) * rstd * g + b, written as the one-kernel form does: same rounding sequence
    const float* xr = x + row * ld;
    float* orow = out + row * ld;
    const RowSplit rs = row_split(row, T, ld, nseg);
    auto f = [&](float v) { return rn_act<ACT>((v - m.mean) * rstd * g + b, act); };
    const long b0 = (long)seg * rs.q4, b1 = lmin(b0 + rs.q4, rs.nb4);
    const float4* body = reinterpret_cast<const float4*>(xr + rs.head);
    float4* obody = reinterpret_cast<float4*>(orow + rs.head);
    for (long i = b0 + tid; i < b1; i += 256) {
        const float4 v = body[i];
        obody[i] = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
    }
    if (seg == 0 && tid < rs.head) orow[tid] = f(xr[tid]);
    if (seg == nseg - 1 && tid >= 64 && tid - 64 < rs.tail) {
        const long i = rs.head + 4 * rs.nb4 + (tid - 64);
        orow[i] = f(xr[i]);
    }
}

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_layernorm_ct(const float* x, const float* res, const float* gamma, const float* beta, float* out, int N,
                                 int C, int64_t T, float eps, int64_t x_sn, int64_t r_sn, int64_t o_sn, void* stream) {
    if (!x || !gamma || !beta || !out) return fail(AICG_E_ARG, "aicg_layernorm_ct: null pointer");
    if (N < 0 || C < 1 || T < 0) return fail(AICG_E_SHAPE, "aicg_layernorm_ct: bad shape");
    if (N == 0 || T == 0) return AICG_OK;
    if (C <= 1024) {
        hipStream_t st = (hipStream_t)stream;
        // 32-column blocks (128-byte segments) from half a wave of workgroups per CU on: T = 13198 (HuBERT batched over a rank's chunks)
        // 79 -> 46 us = 2.7 TB/s with the residual, (192, 6600) 13.8 -> 9.7 us; below that the 8-column form's 4x workgroups win
        const bool narrow = ldiv_up(T, 32) * N < 128;
        dim3 grid((unsigned)ldiv_up(T, narrow ? 8 : 32), (unsigned)N);
#define AICG_LN_LAUNCH(NV, CB) hipLaunchKernelGGL(HIP_KERNEL_NAME(layernorm_ct_reg_kernel<NV, CB>), grid, dim3(1024), 0, st, x, res, gamma, \
                                                  beta, out, C, (long)T, eps, (long)x_sn, (long)r_sn, (long)o_sn)
        if (narrow) {       // 128 channel groups
            if (C <= 256) AICG_LN_LAUNCH(2, 8);
            else if (C <= 768) AICG_LN_LAUNCH(6, 8);
            else AICG_LN_LAUNCH(8, 8);
        } else {            // 32 channel groups
            if (C <= 256) AICG_LN_LAUNCH(8, 32);
            else if (C <= 768) AICG_LN_LAUNCH(24, 32);
            else AICG_LN_LAUNCH(32, 32);
        }
#undef AICG_LN_LAUNCH
        return check_launch("layernorm_ct_reg_kernel");
    }
    dim3 grid((unsigned)ldiv_up(T, 64), (unsigned)N);
    hipLaunchKernelGGL(layernorm_ct_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, out, C, (long)T,
                       eps, (long)x_sn, (long)r_sn, (long)o_sn);
    return check_launch("layernorm_ct_kernel");
}

extern "C" int aicg_rownorm_act_workspace_floats(int rows, int64_t T, int64_t* n_floats) {
    if (!n_floats) return fail(AICG_E_ARG, "aicg_rownorm_act_workspace_floats: null pointer");
    *n_floats = (rows <= 0 || T < 4096) ? 0 : 3L * rows * ldiv_up(T, 4L * 256 * kRnV);   // short rows: the one-kernel form
    return AICG_OK;
}

extern "C" int aicg_rownorm_act_ld(const float* x, const float* gamma, const float* beta, float* out, int rows, int64_t T, int64_t ld,
                                   float eps, int act, float* workspace, void* stream) {
    if (!x || !out) return fail(AICG_E_ARG, "aicg_rownorm_act: null pointer");
    if (rows < 0 || T < 1 || ld < T) return fail(AICG_E_SHAPE, "aicg_rownorm_act: bad shape (rows %d, T %ld, row stride %ld)", rows, (long)T, (long)ld);
    if (rows == 0) return AICG_OK;
    if (workspace && T >= 4096 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        const int nseg = (int)ldiv_up(T, 4L * 256 * kRnV);
        dim3 grid((unsigned)nseg, (unsigned)rows);
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(rownorm_stats_kernel, grid, dim3(256), 0, st, x, workspace, (long)T, (long)ld, nseg);
        if (act == AICG_ACT_GELU)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(rownorm_apply_kernel<AICG_ACT_GELU>), grid, dim3(256), 0, st, x, (const float*)workspace,
                               gamma, beta, out, (long)T, (long)ld, nseg, eps, act);
        else if (act == AICG_ACT_NONE)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(rownorm_apply_kernel<AICG_ACT_NONE>), grid, dim3(256), 0, st, x, (const float*)workspace,
                               gamma, beta, out, (long)T, (long)ld, nseg, eps, act);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(rownorm_apply_kernel<-1>), grid, dim3(256), 0, st, x, (const float*)workspace, gamma,
                               beta, out, (long)T, (long)ld, nseg, eps, act);
        return check_launch("rownorm_apply_kernel");
    }
    hipLaunchKernelGGL(rownorm_act_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out,
                       (long)T, (long)ld, eps, act);
    return check_launch("rownorm_act_kernel");
}

extern "C" int aicg_rownorm_act(const float* x, const float* gamma, const float* beta, float* out, int rows, int64_t T,
                                float eps, int act, float* workspace, void* stream) {
    return aicg_rownorm_act_ld(x, gamma, beta, out, rows, T, T, eps, act, workspace, stream);
}
