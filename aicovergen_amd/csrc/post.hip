// Host-side pre/post steps of VC.pipeline moved onto the device:
//   * quietest-sample cut search: 160-tap box sum + argmin |.| (reference src/vc_infer_pipeline.py:516-528)
//   * change_rms: librosa.feature.rms envelopes, linear interpolation, power-law mix (:41-60)
//   * peak limit + truncating int16 conversion (:645-649)
// All HBM-bound single passes; float64 where the reference computes in float64 (filtered audio, box sums).
#include "common.h"

namespace aicg {

// out[j] = sum_{i=0}^{win-1} x[j + i], accumulated in the reference's order (i ascending, starting from 0.0):
// bit-equal to `for i in range(window): audio_sum += audio_pad[i : i - window]`.
__global__ void __launch_bounds__(256) box_sum_f64_kernel(const double* __restrict__ x, double* __restrict__ out, long n, int win) {
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < win; ++i) s += x[j + i];
        out[j] = s;
    }
}

// index of the first minimum of |x[start .. start+len)| ; one workgroup per segment
__global__ void __launch_bounds__(256) argmin_abs_f64_kernel(const double* __restrict__ x, const long* __restrict__ starts,
                                                             const long* __restrict__ lens, long* __restrict__ out) {
    __shared__ double bv[256];
    __shared__ long bi[256];
    const long s0 = starts[blockIdx.x], len = lens[blockIdx.x];
    double best = INFINITY;
    long bidx = 0x7fffffffffffffffL;
    for (long i = threadIdx.x; i < len; i += 256) {
        const double v = fabs(x[s0 + i]);
        if (v < best || (v == best && i < bidx)) { best = v; bidx = i; }
    }
    bv[threadIdx.x] = best;
    bi[threadIdx.x] = bidx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double ov = bv[threadIdx.x + o];
            const long oi = bi[threadIdx.x + o];
            if (ov < bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = ov; bi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = bi[0];
}

__device__ __forceinline__ long reflect_index(long p, long pad, long n) {
    long i = p - pad;               // np.pad(mode="reflect"): ... x2 x1 | x0 x1 ... x(n-1) | x(n-2) ...
    if (n == 1) return 0;
    const long period = 2 * (n - 1);  // repeated reflection (pad longer than the signal: clips of <= 0.5 s) has this period
    if (i < 0 || i >= period) { i %= period; if (i < 0) i += period; }
    if (i >= n) i = period - i;
    return i;
}

// librosa.feature.rms(y, frame_length, hop_length) (center=True, reflect): one workgroup per frame
template <typename T>
__global__ void __launch_bounds__(256) frame_rms_kernel(const T* __restrict__ x, double* __restrict__ out, long n, int frame, int hop) {
    __shared__ double sh[4];
    const long f = blockIdx.x;
    const long pad = frame / 2;
    double s = 0.0;
    for (int i = threadIdx.x; i < frame; i += 256) {
        const double v = (double)x[reflect_index(f * hop + i, pad, n)];
        s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[f] = sqrt((sh[0] + sh[1] + sh[2] + sh[3]) / (double)frame);
}

// F.interpolate(mode="linear", align_corners=False) of an envelope of `m` points to position i of n
__device__ __forceinline__ double interp_env(const double* __restrict__ e, long m, long i, long n) {
    double src = ((double)i + 0.5) * ((double)m / (double)n) - 0.5;
    if (src < 0.0) src = 0.0;
    long i0 = (long)src;
    if (i0 > m - 1) i0 = m - 1;
    const long i1 = i0 + 1 < m ? i0 + 1 : m - 1;
    const double w = src - (double)i0;
    return (1.0 - w) * e[i0] + w * e[i1];
}

// data2 *= rms1^(1-rate) * max(rms2, 1e-6)^(rate-1)
__global__ void __launch_bounds__(256) rms_mix_kernel(float* __restrict__ data, long n, const double* __restrict__ rms1, long m1,
                                                      const double* __restrict__ rms2, long m2, double rate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double r1 = interp_env(rms1, m1, i, n);
        double r2 = interp_env(rms2, m2, i, n);
        if (r2 < 1e-6) r2 = 1e-6;
        data[i] = (float)((double)data[i] * (pow(r1, 1.0 - rate) * pow(r2, rate - 1.0)));
    }
}

__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // non-negative floats order like their bit patterns: one integer atomic max per wave
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, (unsigned)__float_as_int(m));
}

// (x * scale).astype(int16): C-style truncation toward zero
__global__ void __launch_bounds__(256) to_int16_kernel(const float* __restrict__ x, short* __restrict__ out, long n, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (short)(int)(x[i] * scale);
}

static unsigned pgrid(long total) { return (unsigned)lmax(1, lmin((total + 255) / 256, 256L * 16)); }

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_box_sum_f64(const double* x, double* out, int64_t n, int window, void* stream) {
    if (!x || !out) return fail(AICG_E_ARG, "aicg_box_sum_f64: null pointer");
    if (n <= 0) return AICG_OK;
    hipLaunchKernelGGL(box_sum_f64_kernel, dim3(pgrid(n)), dim3(256), 0, (hipStream_t)stream, x, out, (long)n, window);
    return check_launch("box_sum_f64_kernel");
}

extern "C" int aicg_argmin_abs_f64(const double* x, const int64_t* starts, const int64_t* lens, int64_t* out, int n_seg,
                                   void* stream) {
    if (!x || !starts || !lens || !out) return fail(AICG_E_ARG, "aicg_argmin_abs_f64: null pointer");
    if (n_seg <= 0) return AICG_OK;
    hipLaunchKernelGGL(argmin_abs_f64_kernel, dim3((unsigned)n_seg), dim3(256), 0, (hipStream_t)stream, x, (const long*)starts,
                       (const long*)lens, (long*)out);
    return check_launch("argmin_abs_f64_kernel");
}

extern "C" int aicg_frame_rms(const void* x, int is_f64, double* out, int64_t n, int frame_length, int hop_length, void* stream) {
    if (!x || !out) return fail(AICG_E_ARG, "aicg_frame_rms: null pointer");
    if (n <= 0 || frame_length <= 0 || hop_length <= 0) return fail(AICG_E_SHAPE, "aicg_frame_rms: empty signal or bad framing");
    const long n_frames = 1 + n / hop_length;  // 1 + (n + 2*(frame/2) - frame) / hop with an even frame
    if (is_f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(frame_rms_kernel<double>), dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream,
                           (const double*)x, out, (long)n, frame_length, hop_length);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(frame_rms_kernel<float>), dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, out, (long)n, frame_length, hop_length);
    return check_launch("frame_rms_kernel");
}

extern "C" int aicg_rms_mix(float* data, int64_t n, const double* rms1, int64_t m1, const double* rms2, int64_t m2, double rate,
                            void* stream) {
    if (!data || !rms1 || !rms2) return fail(AICG_E_ARG, "aicg_rms_mix: null pointer");
    if (n <= 0) return AICG_OK;
    hipLaunchKernelGGL(rms_mix_kernel, dim3(pgrid(n)), dim3(256), 0, (hipStream_t)stream, data, (long)n, rms1, (long)m1, rms2,
                       (long)m2, rate);
    return check_launch("rms_mix_kernel");
}

extern "C" int aicg_absmax(const float* x, int64_t n, float* out, void* stream) {
    if (!x || !out) return fail(AICG_E_ARG, "aicg_absmax: null pointer");
    (void)hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream);
    if (n <= 0) return AICG_OK;
    hipLaunchKernelGGL(absmax_kernel, dim3(pgrid(n)), dim3(256), 0, (hipStream_t)stream, x, (long)n, (unsigned*)out);
    return check_launch("absmax_kernel");
}

extern "C" int aicg_to_int16(const float* x, int16_t* out, int64_t n, float scale, void* stream) {
    if (!x || !out) return fail(AICG_E_ARG, "aicg_to_int16: null pointer");
    if (n <= 0) return AICG_OK;
    hipLaunchKernelGGL(to_int16_kernel, dim3(pgrid(n)), dim3(256), 0, (hipStream_t)stream, x, (short*)out, (long)n, scale);
    return check_launch("to_int16_kernel");
}
