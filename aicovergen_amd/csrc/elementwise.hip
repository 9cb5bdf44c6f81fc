// HBM-bound helper kernels of the RVC synthesizer: col2im for transposed convolutions, the NSF sine
// source, the WaveNet gate, the prior sample, nearest x2 + protect blend.  One pass over the data each,
// unit-stride along time so every wave issues full 256-byte rows.
#include "common.h"

#include <cstdint>

namespace aicg {

// ---- col2im: gather form of ConvTranspose (reference models.py:453-463 ups; rmvpe.py:147-155) ---------
// cols: (N, Cout*KH*KW, Hi, Wi) = the 1x1 GEMM  W[ci][co,kh,kw]^T x ;  out[n,co,ho,wo] =
//   act(bias[co] + sum over (kh,kw) with (ho+ph-kh) % sh == 0, (wo+pw-kw) % sw == 0 of cols[n,(co,kh,kw),hi,wi]) + add
struct Col2imArgs {
    const float* cols;
    const float* bias;
    const float* add;
    float* out;
    int N, Cout, Hi, Wi, Ho, Wo, KH, KW, sh, sw, ph, pw, act;
    float slope;
    long o_sn, o_sc, o_sh, a_sn, a_sc, a_sh;
};

__global__ void __launch_bounds__(256) col2im_kernel(Col2imArgs p) {
    const long total = (long)p.N * p.Cout * p.Ho * p.Wo;
    const long plane = (long)p.Hi * p.Wi;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int wo = (int)(i % p.Wo);
        long t = i / p.Wo;
        const int ho = (int)(t % p.Ho);
        t /= p.Ho;
        const int co = (int)(t % p.Cout);
        const int n = (int)(t / p.Cout);
        float acc = p.bias ? p.bias[co] : 0.f;
        const float* cn = p.cols + (long)n * p.Cout * p.KH * p.KW * plane;
        for (int kh = (ho + p.ph) % p.sh; kh < p.KH; kh += p.sh) {
            const int hi = (ho + p.ph - kh) / p.sh;
            if (ho + p.ph - kh < 0) break;
            if (hi >= p.Hi) continue;
            for (int kw = (wo + p.pw) % p.sw; kw < p.KW; kw += p.sw) {
                const int wi = (wo + p.pw - kw) / p.sw;
                if (wo + p.pw - kw < 0) break;
                if (wi >= p.Wi) continue;
                acc += cn[((long)(co * p.KH + kh) * p.KW + kw) * plane + (long)hi * p.Wi + wi];
            }
        }
        acc = apply_act(acc, p.act, p.slope);
        if (p.add) acc += p.add[(long)n * p.a_sn + (long)co * p.a_sc + (long)ho * p.a_sh + wo];
        p.out[(long)n * p.o_sn + (long)co * p.o_sc + (long)ho * p.o_sh + wo] = acc;
    }
}

// ---- col2im, one-dimensional layers: the vocoder's up-sampling stages (reference models.py:494-505: ConvTranspose1d k 16 / s 10, k 4 / s 2, ...)
// The generic kernel above walks OUTPUT elements (two 64-bit divisions each, and for stride s only every s-th lane reads the same
// row of `cols`).  Here a thread owns one INPUT position q of one channel and forms the s outputs s q - pad + r, r = 0 .. s - 1:
//     out[s q + r - pad] = bias + sum_j cols[co K + r + s j][q - j]            (j = 0 .. ceil(K / s) - 1, taps r + s j < K)
// -- every row of `cols` is read as one unit-stride stream, 32-bit index arithmetic, no division.  The s x 256 outputs of a workgroup
// are contiguous in the output row: they pass through LDS and leave as one unit-stride stream too (together with the `add` operand).
template <int S>
__global__ void __launch_bounds__(256) col2im1d_kernel(Col2imArgs p, int J) {
    HIP_DYNAMIC_SHARED(float, c2i_tile)                     // 256 * s floats
    const int s = S > 0 ? S : p.sw;
    const int co = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
    const int q0 = blockIdx.x * 256, q = q0 + tid;
    const float* cn = p.cols + ((long)n * p.Cout + co) * p.KW * (long)p.Wi;
    const float b = p.bias ? p.bias[co] : 0.f;
#pragma unroll 2
    for (int r = 0; r < s; ++r) {
        float acc = b;
        for (int j = 0; j < J; ++j) {
            const int kk = r + s * j, qi = q - j;
            if (kk < p.KW && qi >= 0 && qi < p.Wi) acc += cn[(long)kk * p.Wi + qi];
        }
        c2i_tile[tid * s + r] = apply_act(acc, p.act, p.slope);
    }
    __syncthreads();
    const long nbase = (long)s * q0 - p.pw;                 // output index of the tile's first element
    float* orow = p.out + (long)n * p.o_sn + (long)co * p.o_sc;
    const float* arow = p.add ? p.add + (long)n * p.a_sn + (long)co * p.a_sc : nullptr;
    for (int i = tid; i < 256 * s; i += 256) {
        const long no = nbase + i;
        if (no >= 0 && no < p.Wo) orow[no] = c2i_tile[i] + (arow ? arow[no] : 0.f);
    }
}

// ---- NSF sine source (reference models.py:320-370 SineGen.forward + :414-419 SourceModuleHnNSF) --------
// harmonic_num = 0.  phase[n] = sum_{m<=n} rad[floor(m/upp)], rad = (f0/sr) mod 1; the reference's
// cumsum_shift only removes integers from the running fp32 sum, so the phase is
// carried modulo 1 in fp64: frame prefix (one thread block scan) + (i+1)*rad inside the frame.
__global__ void __launch_bounds__(256) sine_frame_prefix_kernel(const float* __restrict__ f0, double* __restrict__ prefix,
                                                                int T, int upp, float sr) {
    // single workgroup: blocked scan over frames; prefix[t] = frac(sum_{u<t} rad_u * upp).  The 256 per-thread partial sums are
    // combined by thread 0 in index order (the fp64 "add, drop the integer part" chain is order dependent: a tree would round
    // differently from the sequential reference order the tests pin), from registers of one wave-wide read -- 256 dependent fp64 steps,
    // ~4 us; the per-thread parts (T / 256 frames each) run in parallel before and after it.
    __shared__ double part[256];
    const int tid = threadIdx.x;
    const int per = (T + 255) / 256;
    const int t0 = tid * per, t1 = imin(T, t0 + per);
    double s = 0.0;
    for (int t = t0; t < t1; ++t) {
        const float rad = fmodf(f0[t] / sr, 1.0f);
        s += (double)rad * (double)upp;
        s -= floor(s);
    }
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) {
            const double v = part[i];
            part[i] = run;
            run += v;
            run -= floor(run);
        }
    }
    __syncthreads();
    double run = part[tid];
    for (int t = t0; t < t1; ++t) {
        prefix[t] = run;
        const float rad = fmodf(f0[t] / sr, 1.0f);
        run += (double)rad * (double)upp;
        run -= floor(run);
    }
}

// One thread = FOUR consecutive samples of one frame (upp % 4 == 0: 400 / 480 / 320 of the 40k / 48k / 32k models): one float4 of
// noise in, one float4 out, the frame's f0 / prefix / rad read once.  (r3: one sample per thread, scalar loads and stores, 3 % of the
// HBM rate; what remains of the time is the two launches and the scan above -- 21 MB per chunk are a 4 us stream.)
__global__ void __launch_bounds__(256) sine_source4_kernel(const float* __restrict__ f0, const double* __restrict__ prefix,
                                                           const float* __restrict__ noise, float* __restrict__ out,
                                                           int T, int upp, float sr, float sine_amp, float noise_std,
                                                           float lin_w, float lin_b) {
    const int q_per = upp >> 2;
    const long total4 = (long)T * q_per;
    for (long n4 = (long)blockIdx.x * blockDim.x + threadIdx.x; n4 < total4; n4 += (long)gridDim.x * blockDim.x) {
        const int t = (int)(n4 / q_per);
        const int i0 = (int)(n4 - (long)t * q_per) * 4;
        const float f = f0[t];
        const float rad = fmodf(f / sr, 1.0f);
        const double base = prefix[t];
        const float uv = f > 0.f ? 1.f : 0.f;
        const float namp = uv * noise_std + (1.f - uv) * sine_amp / 3.f;
        const float4 nz = *reinterpret_cast<const float4*>(noise + (long)t * upp + i0);
        const float nv[4] = {nz.x, nz.y, nz.z, nz.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double ph = base + (double)(i0 + j + 1) * (double)rad;
            ph -= floor(ph);
            const float sine = sinf((float)(ph * 6.283185307179586476925)) * sine_amp;
            const float v = sine * uv + namp * nv[j];
            o[j] = tanhf(lin_w * v + lin_b);  // l_linear (1->1) + tanh (models.py:418)
        }
        *reinterpret_cast<float4*>(out + (long)t * upp + i0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ void __launch_bounds__(256) sine_source_kernel(const float* __restrict__ f0, const double* __restrict__ prefix,
                                                          const float* __restrict__ noise, float* __restrict__ out,
                                                          int T, int upp, float sr, float sine_amp, float noise_std,
                                                          float lin_w, float lin_b) {
    const long total = (long)T * upp;
    for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (long)gridDim.x * blockDim.x) {
        const int t = (int)(n / upp);
        const int i = (int)(n - (long)t * upp);
        const float f = f0[t];
        const float rad = fmodf(f / sr, 1.0f);
        double ph = prefix[t] + (double)(i + 1) * (double)rad;
        ph -= floor(ph);
        const float sine = sinf((float)(ph * 6.283185307179586476925)) * sine_amp;
        const float uv = f > 0.f ? 1.f : 0.f;
        const float namp = uv * noise_std + (1.f - uv) * sine_amp / 3.f;
        const float v = sine * uv + namp * noise[n];
        out[n] = tanhf(lin_w * v + lin_b);  // l_linear (1->1) + tanh (models.py:418)
    }
}

// ---- WaveNet gate: out[c] = tanh(a[c]) * sigmoid(a[c + C])  (commons.py:105-112; bias/cond already added) -----
__global__ void __launch_bounds__(256) gate_kernel(const float* __restrict__ a, float* __restrict__ out, int N, int C,
                                                   long T) {
    const long total = (long)N * C * T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long t = i % T;
        const long nc = i / T;
        const int c = (int)(nc % C);
        const long n = nc / C;
        const float ta = a[(n * 2 * C + c) * T + t];
        const float sa = a[(n * 2 * C + C + c) * T + t];
        out[i] = tanhf(ta) * (1.f / (1.f + expf(-sa)));
    }
}

// ---- prior sample: z_p = m + exp(logs) * noise * scale  (models.py:748); stats = [m ; logs] -----------------
__global__ void __launch_bounds__(256) prior_sample_kernel(const float* __restrict__ stats, const float* __restrict__ noise,
                                                           float* __restrict__ out, int C, long T, float scale) {
    const long total = (long)C * T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float m = stats[i], logs = stats[total + i];
        out[i] = m + expf(logs) * noise[i] * scale;
    }
}

// ---- nearest x2 upsample + protect blend + transpose to channel-major  (vc_infer_pipeline.py:433-452) --------
// feats: (Th, C) token-major HuBERT output.  out[c][t] = f*pf + f*(1-pf) semantics of the reference:
//   feats = feats * pitchff + feats0 * (1 - pitchff), where feats0 is the pre-index copy (== feats when no
//   faiss index is used) and pitchff = 1 if pitchf > 0 else protect.   feats0 may be null (no protect).
__global__ void __launch_bounds__(256) feats_prepare_kernel(const float* __restrict__ feats, const float* __restrict__ feats0,
                                                            const float* __restrict__ pitchf, float* __restrict__ out,
                                                            int Th, int C, int T, float protect) {
    // tile transpose through LDS: reads unit-stride along C, writes unit-stride along T
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        float v = 0.f;
        if (t < T && c < C) {
            const int th = imin(t >> 1, Th - 1);
            v = feats[(long)th * C + c];
            if (feats0) {
                const float f = pitchf[t];
                float pf = f;
                if (f > 0.f) pf = 1.f;      // pitchff[pitchf > 0] = 1
                if (f < 1.f) pf = protect;  // pitchff[pitchf < 1] = protect  (both tests on the original pitchf)
                v = v * pf + feats0[(long)th * C + c] * (1.f - pf);
            }
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        if (c < C && t < T) out[(long)c * T + t] = tile[tx][r];
    }
}

static unsigned ew_grid(long total) { return (unsigned)lmax(1, lmin((total + 255) / 256, 256L * 16)); }

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_col2im(const float* cols, const float* bias, const float* add, float* out, int N, int Cout, int Hi,
                           int Wi, int Ho, int Wo, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w,
                           int act, float slope, int64_t o_sn, int64_t o_sc, int64_t o_sh, int64_t a_sn, int64_t a_sc,
                           int64_t a_sh, void* stream) {
    if (!cols || !out) return fail(AICG_E_ARG, "aicg_col2im: null pointer");
    if (KH < 1 || KW < 1 || stride_h < 1 || stride_w < 1) return fail(AICG_E_SHAPE, "aicg_col2im: bad geometry");
    const long total = (long)N * Cout * Ho * Wo;
    if (total == 0) return AICG_OK;
    Col2imArgs p{cols, bias, add, out, N, Cout, Hi, Wi, Ho, Wo, KH, KW, stride_h, stride_w, pad_h, pad_w, act, slope,
                 (long)o_sn, (long)o_sc, (long)o_sh, (long)a_sn, (long)a_sc, (long)a_sh};
    if (Hi == 1 && Ho == 1 && KH == 1 && stride_h == 1 && pad_h == 0 && Cout <= 65535 && N <= 65535 && stride_w <= 32) {
        // one-dimensional layer: the input-centric form.  Positions q = 0 .. ceil((Wo + pad) / s): the last output is s q + r - pad
        const int J = (KW + stride_w - 1) / stride_w;
        const long nq = ((long)Wo + pad_w + stride_w - 1) / stride_w + 1;
        dim3 grid((unsigned)ldiv_up(nq, 256), (unsigned)Cout, (unsigned)N);
        const size_t lds = (size_t)256 * stride_w * sizeof(float);
        if (stride_w == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(col2im1d_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, p, J);
        else if (stride_w == 10) hipLaunchKernelGGL(HIP_KERNEL_NAME(col2im1d_kernel<10>), grid, dim3(256), lds, (hipStream_t)stream, p, J);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(col2im1d_kernel<0>), grid, dim3(256), lds, (hipStream_t)stream, p, J);
        return check_launch("col2im1d_kernel");
    }
    hipLaunchKernelGGL(col2im_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("col2im_kernel");
}

extern "C" int aicg_sine_source(const float* f0, const float* noise, double* prefix_scratch, float* out, int T, int upp,
                                float sr, float sine_amp, float noise_std, float lin_w, float lin_b, void* stream) {
    if (!f0 || !noise || !prefix_scratch || !out) return fail(AICG_E_ARG, "aicg_sine_source: null pointer");
    if (T < 0 || upp < 1) return fail(AICG_E_SHAPE, "aicg_sine_source: bad shape");
    if (T == 0) return AICG_OK;
    hipLaunchKernelGGL(sine_frame_prefix_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, f0, prefix_scratch, T, upp, sr);
    if ((upp & 3) == 0 && ((uintptr_t)noise & 15) == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(sine_source4_kernel, dim3(ew_grid((long)T * (upp >> 2))), dim3(256), 0, (hipStream_t)stream, f0,
                           (const double*)prefix_scratch, noise, out, T, upp, sr, sine_amp, noise_std, lin_w, lin_b);
    else
        hipLaunchKernelGGL(sine_source_kernel, dim3(ew_grid((long)T * upp)), dim3(256), 0, (hipStream_t)stream, f0,
                           (const double*)prefix_scratch, noise, out, T, upp, sr, sine_amp, noise_std, lin_w, lin_b);
    return check_launch("sine_source_kernel");
}

extern "C" int aicg_gate_tanh_sigmoid(const float* a, float* out, int N, int C, int64_t T, void* stream) {
    if (!a || !out) return fail(AICG_E_ARG, "aicg_gate_tanh_sigmoid: null pointer");
    const long total = (long)N * C * T;
    if (total == 0) return AICG_OK;
    hipLaunchKernelGGL(gate_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, a, out, N, C, (long)T);
    return check_launch("gate_kernel");
}

extern "C" int aicg_prior_sample(const float* stats, const float* noise, float* out, int C, int64_t T, float scale,
                                 void* stream) {
    if (!stats || !noise || !out) return fail(AICG_E_ARG, "aicg_prior_sample: null pointer");
    if ((long)C * T == 0) return AICG_OK;
    hipLaunchKernelGGL(prior_sample_kernel, dim3(ew_grid((long)C * T)), dim3(256), 0, (hipStream_t)stream, stats, noise,
                       out, C, (long)T, scale);
    return check_launch("prior_sample_kernel");
}

extern "C" int aicg_feats_prepare(const float* feats, const float* feats0, const float* pitchf, float* out, int Th, int C,
                                  int T, float protect, void* stream) {
    if (!feats || !out || (feats0 && !pitchf)) return fail(AICG_E_ARG, "aicg_feats_prepare: null pointer");
    if (T > 2 * Th) return fail(AICG_E_SHAPE, "aicg_feats_prepare: T=%d exceeds 2*Th=%d", T, 2 * Th);
    if (T == 0 || C == 0) return AICG_OK;
    dim3 grid((unsigned)idiv_up(T, 32), (unsigned)idiv_up(C, 32));
    hipLaunchKernelGGL(feats_prepare_kernel, grid, dim3(256), 0, (hipStream_t)stream, feats, feats0, pitchf, out, Th, C, T,
                       protect);
    return check_launch("feats_prepare_kernel");
}

namespace aicg {
__global__ void __launch_bounds__(256) mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}
__global__ void __launch_bounds__(256) axpbypcz_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b,
                                                       float beta, const float* __restrict__ c, float gamma,
                                                       float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = alpha * a[i];
        if (b) v += beta * b[i];
        if (c) v += gamma * c[i];
        out[i] = v;
    }
}
}  // namespace aicg

extern "C" int aicg_mul(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out) return aicg::fail(AICG_E_ARG, "aicg_mul: null pointer");
    if (n == 0) return AICG_OK;
    hipLaunchKernelGGL(aicg::mul_kernel, dim3(aicg::ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n);
    return aicg::check_launch("mul_kernel");
}

extern "C" int aicg_axpbypcz(const float* a, float alpha, const float* b, float beta, const float* c, float gamma, float* out,
                             int64_t n, void* stream) {
    if (!a || !out) return aicg::fail(AICG_E_ARG, "aicg_axpbypcz: null pointer");
    if (n == 0) return AICG_OK;
    hipLaunchKernelGGL(aicg::axpbypcz_kernel, dim3(aicg::ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, alpha, b, beta, c,
                       gamma, out, (long)n);
    return aicg::check_launch("axpbypcz_kernel");
}

// ---- CREPE helpers (torchcrepe 0.0.20 as called at reference src/vc_infer_pipeline.py:116-126) ----------------------
namespace aicg {
// frames -= mean; frames /= max(1e-10, std (unbiased)) per 1024-sample frame (torchcrepe.preprocess)
__global__ void __launch_bounds__(256) frame_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, int L) {
    __shared__ float sh[4];
    const float* xr = x + (long)blockIdx.x * L;
    float s = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) s += xr[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)L;
    __syncthreads();
    float v = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) { const float d = xr[i] - mean; v += d * d; }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float sd = fmaxf(1e-10f, sqrtf((sh[0] + sh[1] + sh[2] + sh[3]) / (float)(L - 1)));
    float* orow = out + (long)blockIdx.x * L;
    for (int i = threadIdx.x; i < L; i += 256) orow[i] = (xr[i] - mean) / sd;
}

// eval BatchNorm (per-channel affine) followed by MaxPool (2,1)/(2,1) along the last axis: (N, C, W) -> (N, C, W/2)
__global__ void __launch_bounds__(256) affine_maxpool2_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ out, int C,
                                                              int Wo, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i / Wo) % C);
        const float s = scale[c], t = shift[c];
        const float a = x[2 * i] * s + t, b = x[2 * i + 1] * s + t;
        out[i] = fmaxf(a, b);
    }
}
}  // namespace aicg

extern "C" int aicg_frame_normalize(const float* x, float* out, int64_t n_frames, int frame_len, void* stream) {
    if (!x || !out) return aicg::fail(AICG_E_ARG, "aicg_frame_normalize: null pointer");
    if (frame_len < 2) return aicg::fail(AICG_E_SHAPE, "aicg_frame_normalize: frame too short");
    if (n_frames <= 0) return AICG_OK;
    hipLaunchKernelGGL(aicg::frame_normalize_kernel, dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, x, out, frame_len);
    return aicg::check_launch("frame_normalize_kernel");
}

extern "C" int aicg_affine_maxpool2(const float* x, const float* scale, const float* shift, float* out, int N, int C, int W,
                                    void* stream) {
    if (!x || !scale || !shift || !out) return aicg::fail(AICG_E_ARG, "aicg_affine_maxpool2: null pointer");
    if (W & 1) return aicg::fail(AICG_E_SHAPE, "aicg_affine_maxpool2: W must be even");
    const long total = (long)N * C * (W / 2);
    if (total == 0) return AICG_OK;
    hipLaunchKernelGGL(aicg::affine_maxpool2_kernel, dim3(aicg::ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                       out, C, W / 2, total);
    return aicg::check_launch("affine_maxpool2_kernel");
}
