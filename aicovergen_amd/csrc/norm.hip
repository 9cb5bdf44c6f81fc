// Normalisation kernels (HBM-bound, two-pass statistics in fp32 like torch):
//   * layernorm_ct : LayerNorm over channels of a channel-major (C, T) map, optional fused residual add
//                    (reference modules.LayerNorm, src/infer_pack/modules.py:29-32, called from
//                     attentions.Encoder.forward :61-73 as norm(x + y); HuBERT encoder LayerNorms)
//   * rownorm_act  : per-row mean/variance over time + affine + activation: HuBERT feature-extractor
//                    GroupNorm(512 groups == 512 channels) + GELU (fairseq ConvFeatureExtractionModel layer 0)
#include "common.h"

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

// Register-resident form for C <= 32 * NV: block = 32 time columns x 32 channel groups, every element is read once, kept in
// registers for the two-pass mean / variance and written once (the strided form above re-reads it three times from 52 workgroups
// for a HuBERT layer: 183 us for 10 MB).
template <int NV>
__global__ void __launch_bounds__(1024) layernorm_ct_reg_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ out, int C, long T, float eps, long x_sn,
                                                                long r_sn, long o_sn) {
    __shared__ float red[32][33];
    __shared__ float stat[32];
    const int tx = threadIdx.x & 31, gy = threadIdx.x >> 5;
    const long t = (long)blockIdx.x * 32 + tx;
    const int n = blockIdx.y;
    const bool ok = t < T;
    const long tt = ok ? t : 0;
    const float* xn = x + (long)n * x_sn;
    const float* rn = res ? res + (long)n * r_sn : nullptr;
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int c = gy + 32 * e;
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
#pragma unroll
        for (int k = 0; k < 32; ++k) m += red[k][tx];
        stat[tx] = m / (float)C;
    }
    __syncthreads();
    const float mean = stat[tx];
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float d = (gy + 32 * e < C) ? v[e] - mean : 0.f;
        q += d * d;
    }
    __syncthreads();
    red[gy][tx] = q;
    __syncthreads();
    if (gy == 0) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) m += red[k][tx];
        stat[tx] = rsqrtf(m / (float)C + eps);
    }
    __syncthreads();
    const float rstd = stat[tx];
    if (!ok) return;
    float* on = out + (long)n * o_sn;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int c = gy + 32 * e;
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
                                                          float eps, int act) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const float* xr = x + row * T;
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
    float* orow = out + row * T;
    for (long t = threadIdx.x; t < T; t += 256) orow[t] = apply_act((xr[t] - mean) * rstd * g + b, act, 0.f);
}

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_layernorm_ct(const float* x, const float* res, const float* gamma, const float* beta, float* out, int N,
                                 int C, int64_t T, float eps, int64_t x_sn, int64_t r_sn, int64_t o_sn, void* stream) {
    if (!x || !gamma || !beta || !out) return fail(AICG_E_ARG, "aicg_layernorm_ct: null pointer");
    if (N < 0 || C < 1 || T < 0) return fail(AICG_E_SHAPE, "aicg_layernorm_ct: bad shape");
    if (N == 0 || T == 0) return AICG_OK;
    if (C <= 1024) {
        dim3 grid((unsigned)ldiv_up(T, 32), (unsigned)N);
        hipStream_t st = (hipStream_t)stream;
        if (C <= 256)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(layernorm_ct_reg_kernel<8>), grid, dim3(1024), 0, st, x, res, gamma, beta, out, C,
                               (long)T, eps, (long)x_sn, (long)r_sn, (long)o_sn);
        else if (C <= 768)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(layernorm_ct_reg_kernel<24>), grid, dim3(1024), 0, st, x, res, gamma, beta, out, C,
                               (long)T, eps, (long)x_sn, (long)r_sn, (long)o_sn);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(layernorm_ct_reg_kernel<32>), grid, dim3(1024), 0, st, x, res, gamma, beta, out, C,
                               (long)T, eps, (long)x_sn, (long)r_sn, (long)o_sn);
        return check_launch("layernorm_ct_reg_kernel");
    }
    dim3 grid((unsigned)ldiv_up(T, 64), (unsigned)N);
    hipLaunchKernelGGL(layernorm_ct_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, out, C, (long)T,
                       eps, (long)x_sn, (long)r_sn, (long)o_sn);
    return check_launch("layernorm_ct_kernel");
}

extern "C" int aicg_rownorm_act(const float* x, const float* gamma, const float* beta, float* out, int rows, int64_t T,
                                float eps, int act, void* stream) {
    if (!x || !out) return fail(AICG_E_ARG, "aicg_rownorm_act: null pointer");
    if (rows < 0 || T < 1) return fail(AICG_E_SHAPE, "aicg_rownorm_act: bad shape");
    if (rows == 0) return AICG_OK;
    hipLaunchKernelGGL(rownorm_act_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out,
                       (long)T, eps, act);
    return check_launch("rownorm_act_kernel");
}
