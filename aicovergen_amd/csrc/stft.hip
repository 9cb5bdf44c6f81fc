// Framed STFT / iSTFT for gfx950.
//
// Replaces torch.stft / torch.istft at MDXModel.stft/.istft (reference src/mdx.py:37-54) and the STFT
// inside rmvpe.MelSpectrogram (src/rmvpe.py:305-314).
//
// One workgroup transforms FR frames that sit in LDS for the whole transform:
//   load (reflect pad + window, two real samples packed into one complex) -> Stockham radix-{4,2,3,5}
//   passes ping-ponging between two LDS buffers -> real-FFT split step -> strided store.
// HBM sees each input sample once per frame that covers it (L2 absorbs the 7.5x frame overlap) and
// each output bin once: the kernel is HBM/LDS bound, there is no GEMM here (a 7680-point DFT as a
// matrix product would be 600x the flops).
#include "common.h"
#include <cstdint>

#include "fft_reg.h"

namespace aicg {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// tw[i] = exp(-2 pi i * i / M).  INV conjugates it.
template <bool INV>
__device__ __forceinline__ float2 twiddle(const float2* __restrict__ tw, int i) {
    float2 w = tw[i];
    if (INV) w.y = -w.y;
    return w;
}

// One Stockham autosort pass of radix r over nfr frames of M complex points (in -> out).
// Butterfly j (0 <= j < M/r) of a frame reads in[j + t*M/r], multiplies by W_{Ns*r}^{t*(j mod Ns)},
// takes a DFT_r and writes out[(j - j mod Ns)*r + (j mod Ns) + t*Ns].
template <bool INV>
__device__ void fft_pass(const float2* __restrict__ in, float2* __restrict__ out, int M, int r, int Ns,
                         const float2* __restrict__ tw, int nfr, int tid, int nthr) {
    const int T = M / r;
    const int tstep = M / (Ns * r);
    const int total = nfr * T;
    for (int idx = tid; idx < total; idx += nthr) {
        const int f = idx / T;
        const int j = idx - f * T;
        const float2* fin = in + f * M;
        float2* fout = out + f * M;
        const int k = j % Ns;
        const int j0 = (j - k) * r + k;
        if (r == 4) {
            float2 v0 = fin[j], v1 = fin[j + T], v2 = fin[j + 2 * T], v3 = fin[j + 3 * T];
            if (Ns > 1) {
                v1 = cmul(v1, twiddle<INV>(tw, k * tstep));
                v2 = cmul(v2, twiddle<INV>(tw, 2 * k * tstep));
                v3 = cmul(v3, twiddle<INV>(tw, 3 * k * tstep));
            }
            const float2 a = cadd(v0, v2), b = csub(v0, v2), c = cadd(v1, v3), e = csub(v1, v3);
            // forward: d = -i*e ; inverse: d = +i*e
            const float2 d = INV ? make_float2(-e.y, e.x) : make_float2(e.y, -e.x);
            fout[j0] = cadd(a, c);
            fout[j0 + Ns] = cadd(b, d);
            fout[j0 + 2 * Ns] = csub(a, c);
            fout[j0 + 3 * Ns] = csub(b, d);
        } else if (r == 2) {
            float2 v0 = fin[j], v1 = fin[j + T];
            if (Ns > 1) v1 = cmul(v1, twiddle<INV>(tw, k * tstep));
            fout[j0] = cadd(v0, v1);
            fout[j0 + Ns] = csub(v0, v1);
        } else {  // r == 3 or r == 5: direct DFT with roots taken from the same table
            float2 v[5];
            for (int t = 0; t < r; ++t) {
                v[t] = fin[j + t * T];
                if (Ns > 1 && t > 0) v[t] = cmul(v[t], twiddle<INV>(tw, t * k * tstep));
            }
            const int rstep = M / r;
            for (int q = 0; q < r; ++q) {
                float2 acc = v[0];
                for (int t = 1; t < r; ++t) acc = cadd(acc, cmul(v[t], twiddle<INV>(tw, ((q * t) % r) * rstep)));
                fout[j0 + q * Ns] = acc;
            }
        }
    }
}

// Whole transform; `a` holds the input (caller has synchronised), returns the buffer with the result.
template <bool INV>
__device__ float2* fft_lds(float2* a, float2* b, int M, int nfr, const float2* __restrict__ tw, int tid, int nthr) {
    int Ns = 1, rem = M;
    while (rem > 1) {
        const int r = (rem % 4 == 0) ? 4 : (rem % 2 == 0) ? 2 : (rem % 3 == 0) ? 3 : 5;
        fft_pass<INV>(a, b, M, r, Ns, tw, nfr, tid, nthr);
        __syncthreads();
        float2* t = a; a = b; b = t;
        Ns *= r;
        rem /= r;
    }
    return a;
}

struct StftArgs {
    const float* x;
    float* out;
    const float* window;
    const float2* tw_half;
    const float2* tw_full;
    int n_sig, L, n_fft, hop, n_frames, n_bins, fr_per_block;
    long o_sig, o_im, o_bin, o_frame;
};

__global__ void __launch_bounds__(256) stft_kernel(StftArgs p) {
    HIP_DYNAMIC_SHARED(float2, smem)
    const int N = p.n_fft, M = N / 2, FR = p.fr_per_block;
    const int tid = threadIdx.x, nthr = blockDim.x;
    float2* a = smem;
    float2* b = smem + FR * M;
    const long total_frames = (long)p.n_sig * p.n_frames;
    const long fr0 = (long)blockIdx.x * FR;

    for (int idx = tid; idx < FR * M; idx += nthr) {
        const int f = idx / M, m = idx - f * M;
        const long gf = fr0 + f;
        float2 z = make_float2(0.f, 0.f);
        if (gf < total_frames) {
            const int s = (int)(gf / p.n_frames), t = (int)(gf - (long)s * p.n_frames);
            const float* xs = p.x + (long)s * p.L;
            int n0 = t * p.hop - M + 2 * m;  // center=True: frame starts n_fft/2 before t*hop
            int n1 = n0 + 1;
            if (n0 < 0) n0 = -n0;
            if (n1 < 0) n1 = -n1;
            if (n0 >= p.L) n0 = 2 * (p.L - 1) - n0;
            if (n1 >= p.L) n1 = 2 * (p.L - 1) - n1;
            z.x = xs[n0] * p.window[2 * m];
            z.y = xs[n1] * p.window[2 * m + 1];
        }
        a[idx] = z;
    }
    __syncthreads();
    const float2* Z = fft_lds<false>(a, b, M, FR, p.tw_half, tid, nthr);

    // split step: X[k] = E[k] + W_N^k O[k],  E = (Z[k] + conj Z[M-k])/2,  O = (Z[k] - conj Z[M-k])/(2i)
    const int nb = p.n_bins;
    for (int idx = tid; idx < FR * nb; idx += nthr) {
        const int f = idx / nb, k = idx - f * nb;
        const long gf = fr0 + f;
        if (gf >= total_frames) continue;
        const int s = (int)(gf / p.n_frames), t = (int)(gf - (long)s * p.n_frames);
        const float2 zk = Z[f * M + (k == M ? 0 : k)];
        const float2 zm = cconj(Z[f * M + ((k == 0 || k == M) ? 0 : M - k)]);
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y + zm.y));
        const float2 d = csub(zk, zm);
        const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);
        const float2 X = cadd(e, cmul(p.tw_full[k], o));
        float* dst = p.out + (long)s * p.o_sig + (long)k * p.o_bin + (long)t * p.o_frame;
        dst[0] = X.x;
        dst[p.o_im] = X.y;
    }
}

struct IstftArgs {
    const float* spec;
    float* frames;
    const float* window;
    const float2* tw_half;
    const float2* tw_full;
    int n_sig, n_fft, n_frames, n_bins, fr_per_block;
    long i_sig, i_im, i_bin, i_frame;
};

__global__ void __launch_bounds__(256) istft_frames_kernel(IstftArgs p) {
    HIP_DYNAMIC_SHARED(float2, smem)
    const int N = p.n_fft, M = N / 2, FR = p.fr_per_block;
    const int tid = threadIdx.x, nthr = blockDim.x;
    float2* a = smem;
    float2* b = smem + FR * M;
    const long total_frames = (long)p.n_sig * p.n_frames;
    const long fr0 = (long)blockIdx.x * FR;

    // merge step: Z[k] = E[k] + i O[k], E = (X[k] + conj X[M-k])/2, O = (X[k] - conj X[M-k])/2 * conj(W_N^k)
    for (int idx = tid; idx < FR * M; idx += nthr) {
        const int f = idx / M, k = idx - f * M;
        const long gf = fr0 + f;
        float2 z = make_float2(0.f, 0.f);
        if (gf < total_frames) {
            const int s = (int)(gf / p.n_frames), t = (int)(gf - (long)s * p.n_frames);
            const float* base = p.spec + (long)s * p.i_sig + (long)t * p.i_frame;
            const int km = M - k;  // 1..M
            float2 xk = make_float2(0.f, 0.f), xm = make_float2(0.f, 0.f);
            if (k < p.n_bins) { xk.x = base[(long)k * p.i_bin]; xk.y = base[(long)k * p.i_bin + p.i_im]; }
            if (km < p.n_bins) { xm.x = base[(long)km * p.i_bin]; xm.y = base[(long)km * p.i_bin + p.i_im]; }
            if (k == 0) { xk.y = 0.f; xm.y = 0.f; }  // C2R: imaginary parts of DC and Nyquist are ignored
            xm.y = -xm.y;                               // conj X[M-k]
            const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y + xm.y));
            const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y - xm.y));
            const float2 o = cmul(d, cconj(p.tw_full[k]));
            z = make_float2(e.x - o.y, e.y + o.x);
        }
        a[idx] = z;
    }
    __syncthreads();
    const float2* Z = fft_lds<true>(a, b, M, FR, p.tw_half, tid, nthr);
    const float inv = 1.0f / (float)M;
    for (int idx = tid; idx < FR * M; idx += nthr) {
        const int f = idx / M, m = idx - f * M;
        const long gf = fr0 + f;
        if (gf >= total_frames) continue;
        const float2 z = Z[idx];
        float2* dst = reinterpret_cast<float2*>(p.frames + gf * N) + m;
        *dst = make_float2(z.x * inv * p.window[2 * m], z.y * inv * p.window[2 * m + 1]);
    }
}

// out[s][n] = sum_t frames[s][t][n + N/2 - t*hop] / sum_t window^2[n + N/2 - t*hop]
__global__ void __launch_bounds__(256) istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                        float* __restrict__ out, int n_sig, int L, int N, int hop,
                                                        int n_frames) {
    const long total = (long)n_sig * L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int s = (int)(i / L);
        const int n = (int)(i - (long)s * L);
        const int pidx = n + N / 2;
        int t_hi = pidx / hop;
        if (t_hi > n_frames - 1) t_hi = n_frames - 1;
        int t_lo = (pidx - N + hop) / hop;  // ceil((pidx - N + 1)/hop) for pidx-N+1 > 0
        if (pidx - N + 1 <= 0) t_lo = 0;
        float acc = 0.f, env = 0.f;
        const float* fs = frames + (long)s * n_frames * N;
        for (int t = t_lo; t <= t_hi; ++t) {
            const int q = pidx - t * hop;
            const float w = window[q];
            acc += fs[(long)t * N + q];
            env += w * w;
        }
        out[i] = acc / env;
    }
}

// ---- compile-time-plan kernels (fft_reg.h): one workgroup = FR frames, three register passes ------------------------------------------
__device__ __forceinline__ int reflect_index(int n, int L) {
    if (n < 0) n = -n;
    if (n >= L) n = 2 * (L - 1) - n;
    return n;
}

template <class P>
__global__ void __launch_bounds__(P::NT) stft_reg_kernel(StftArgs p) {
    using namespace fftc;
    constexpr int M = P::M, R0 = P::R0, R1 = P::R1, R2 = P::R2;
    HIP_DYNAMIC_SHARED(float2, smem)
    const int fl = threadIdx.x / P::TPF, j = threadIdx.x % P::TPF;
    float2* const buf = smem + fl * P::SLOTS;
    const long gf = (long)blockIdx.x * P::FR + fl;
    const bool live = gf < (long)p.n_sig * p.n_frames;      // uniform over the TPF threads of a frame (whole waves)
    const int s = live ? (int)(gf / p.n_frames) : 0, t = live ? (int)(gf - (long)s * p.n_frames) : 0;
    // pass 0: samples straight from HBM (reflect padding + window), two real samples per complex point
    if (live && j < P::T0) {
        const float* xs = p.x + (long)s * p.L;
        const float2* win = reinterpret_cast<const float2*>(p.window);
        const int n_base = t * p.hop - M;                    // center=True: the frame starts n_fft / 2 before t * hop
        const bool inside = n_base >= 0 && n_base + 2 * M <= p.L;
        float2 v[R0];
#pragma unroll
        for (int u = 0; u < R0; ++u) {
            const int m = j + u * P::T0;
            const int n0 = n_base + 2 * m;
            const float2 w = win[m];
            const float a = xs[inside ? n0 : reflect_index(n0, p.L)], b = xs[inside ? n0 + 1 : reflect_index(n0 + 1, p.L)];
            v[u] = make_float2(a * w.x, b * w.y);
        }
        Dft<R0, false>::run(v);
#pragma unroll
        for (int q = 0; q < R0; ++q) buf[slot(j * R0 + q)] = v[q];
    }
    __syncthreads();
    float2 v1[R1];
    if (live && j < P::T1) load_pass<R1, M, R0, false>(buf, p.tw_half, j, v1);
    __syncthreads();
    if (live && j < P::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) buf[slot(out_pos<R1, R0>(j, q))] = v1[q];
    }
    __syncthreads();
    float2 v2[R2];
    if (live && j < P::T2) load_pass<R2, M, R0 * R1, false>(buf, p.tw_half, j, v2);
    __syncthreads();
    if (live && j < P::T2) {
#pragma unroll
        for (int q = 0; q < R2; ++q) buf[slot(j + q * (R0 * R1))] = v2[q];
    }
    __syncthreads();
    if (!live) return;
    // split step: X[k] = E[k] + W_N^k O[k],  E = (Z[k] + conj Z[M-k])/2,  O = (Z[k] - conj Z[M-k])/(2i)
    float* const dst0 = p.out + (long)s * p.o_sig + (long)t * p.o_frame;
    for (int k = j; k < p.n_bins; k += P::TPF) {
        const float2 zk = buf[slot(k == M ? 0 : k)];
        const float2 zm = cconj(buf[slot((k == 0 || k == M) ? 0 : M - k)]);
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y + zm.y));
        const float2 d = csub(zk, zm);
        const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);
        const float2 X = cadd(e, cmul(p.tw_full[k], o));
        float* dst = dst0 + (long)k * p.o_bin;
        dst[0] = X.x;
        dst[p.o_im] = X.y;
    }
}

template <class P>
__global__ void __launch_bounds__(P::NT) istft_frames_reg_kernel(IstftArgs p) {
    using namespace fftc;
    constexpr int M = P::M, N = 2 * M, R0 = P::R0, R1 = P::R1, R2 = P::R2;
    HIP_DYNAMIC_SHARED(float2, smem)
    const int fl = threadIdx.x / P::TPF, j = threadIdx.x % P::TPF;
    float2* const buf = smem + fl * P::SLOTS;
    const long gf = (long)blockIdx.x * P::FR + fl;
    const bool live = gf < (long)p.n_sig * p.n_frames;
    const int s = live ? (int)(gf / p.n_frames) : 0, t = live ? (int)(gf - (long)s * p.n_frames) : 0;
    // pass 0 input = merge step from HBM: Z[k] = E[k] + i O[k], E = (X[k] + conj X[M-k])/2, O = (X[k] - conj X[M-k])/2 * conj(W_N^k)
    if (live && j < P::T0) {
        const float* base = p.spec + (long)s * p.i_sig + (long)t * p.i_frame;
        float2 v[R0];
#pragma unroll
        for (int u = 0; u < R0; ++u) {
            const int k = j + u * P::T0, km = M - k;
            float2 xk = make_float2(0.f, 0.f), xm = make_float2(0.f, 0.f);
            if (k < p.n_bins) { xk.x = base[(long)k * p.i_bin]; xk.y = base[(long)k * p.i_bin + p.i_im]; }
            if (km < p.n_bins) { xm.x = base[(long)km * p.i_bin]; xm.y = base[(long)km * p.i_bin + p.i_im]; }
            if (k == 0) { xk.y = 0.f; xm.y = 0.f; }  // C2R: imaginary parts of DC and Nyquist are ignored
            xm.y = -xm.y;
            const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y + xm.y));
            const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y - xm.y));
            const float2 o = cmul(d, cconj(p.tw_full[k]));
            v[u] = make_float2(e.x - o.y, e.y + o.x);
        }
        Dft<R0, true>::run(v);
#pragma unroll
        for (int q = 0; q < R0; ++q) buf[slot(j * R0 + q)] = v[q];
    }
    __syncthreads();
    float2 v1[R1];
    if (live && j < P::T1) load_pass<R1, M, R0, true>(buf, p.tw_half, j, v1);
    __syncthreads();
    if (live && j < P::T1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) buf[slot(out_pos<R1, R0>(j, q))] = v1[q];
    }
    __syncthreads();
    if (live && j < P::T2) {
        // last pass: result q of butterfly j is element j + q NS -- lane-contiguous: windowed, scaled, straight to HBM
        float2 v2[R2];
        load_pass<R2, M, R0 * R1, true>(buf, p.tw_half, j, v2);
        const float inv = 1.0f / (float)M;
        const float2* win = reinterpret_cast<const float2*>(p.window);
        float2* dst = reinterpret_cast<float2*>(p.frames + gf * N);
#pragma unroll
        for (int q = 0; q < R2; ++q) {
            const int m = j + q * (R0 * R1);
            const float2 w = win[m];
            dst[m] = make_float2(v2[q].x * inv * w.x, v2[q].y * inv * w.y);
        }
    }
}

// float4 form of istft_ola_kernel: four consecutive output samples share their frame set when L, hop and N / 2 are multiples of 4
__global__ void __launch_bounds__(256) istft_ola4_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                         float* __restrict__ out, int n_sig, int L, int N, int hop, int n_frames) {
    const long total4 = (long)n_sig * (L >> 2);
    for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
        const int s = (int)(i4 / (L >> 2));
        const int n = (int)(i4 - (long)s * (L >> 2)) * 4;
        const int pidx = n + N / 2;
        int t_hi = pidx / hop;
        if (t_hi > n_frames - 1) t_hi = n_frames - 1;
        int t_lo = (pidx - N + hop) / hop;
        if (pidx - N + 1 <= 0) t_lo = 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), env = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* fs = frames + (long)s * n_frames * N;
        for (int t = t_lo; t <= t_hi; ++t) {
            const int q = pidx - t * hop;
            const float4 w = *reinterpret_cast<const float4*>(window + q);
            const float4 f = *reinterpret_cast<const float4*>(fs + (long)t * N + q);
            acc.x += f.x; acc.y += f.y; acc.z += f.z; acc.w += f.w;
            env.x += w.x * w.x; env.y += w.y * w.y; env.z += w.z * w.z; env.w += w.w * w.w;
        }
        *reinterpret_cast<float4*>(out + (long)s * L + n) = make_float4(acc.x / env.x, acc.y / env.y, acc.z / env.z, acc.w / env.w);
    }
}

// the frame lengths of the reference's model table (model_data.json: mdx_n_fft_scale_set 4096 ... 16384) and RMVPE's 1024
using Plan512 = fftc::Plan<8, 8, 8, 4>;
using Plan2048 = fftc::Plan<16, 16, 8, 1>;
using Plan2560 = fftc::Plan<16, 16, 10, 1>;
using Plan3072 = fftc::Plan<16, 16, 12, 1>;
using Plan3840 = fftc::Plan<16, 16, 15, 1>;
using Plan4096 = fftc::Plan<16, 16, 16, 1>;
using Plan8192 = fftc::Plan<16, 16, 32, 1>;

template <class P>
static int launch_stft_reg(const StftArgs& p, hipStream_t st) {
    const size_t lds = (size_t)P::FR * P::SLOTS * sizeof(float2);
    const long total = (long)p.n_sig * p.n_frames;
    if (lds > 64 * 1024) allow_dynamic_lds((const void*)stft_reg_kernel<P>, lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(stft_reg_kernel<P>), dim3((unsigned)ldiv_up(total, P::FR)), dim3(P::NT), lds, st, p);
    return check_launch("stft_reg_kernel");
}

template <class P>
static int launch_istft_reg(const IstftArgs& p, hipStream_t st) {
    const size_t lds = (size_t)P::FR * P::SLOTS * sizeof(float2);
    const long total = (long)p.n_sig * p.n_frames;
    if (lds > 64 * 1024) allow_dynamic_lds((const void*)istft_frames_reg_kernel<P>, lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(istft_frames_reg_kernel<P>), dim3((unsigned)ldiv_up(total, P::FR)), dim3(P::NT), lds, st, p);
    return check_launch("istft_frames_reg_kernel");
}

// 1 = no compile-time plan for this length (the run-time-plan kernels take it)
template <class Args, class F>
static int with_plan(int M, const Args& p, hipStream_t st, F&& launch) {
    (void)p; (void)st;
    switch (M) {
        case 512: return launch(Plan512{});
        case 2048: return launch(Plan2048{});
        case 2560: return launch(Plan2560{});
        case 3072: return launch(Plan3072{});
        case 3840: return launch(Plan3840{});
        case 4096: return launch(Plan4096{});
        case 8192: return launch(Plan8192{});
        default: return 1;
    }
}

static bool smooth235(int m) {
    if (m < 1) return false;
    for (int p : {2, 3, 5})
        while (m % p == 0) m /= p;
    return m == 1;
}

static int pick_frames_per_block(int M) {
    // two ping-pong buffers of FR*M complex floats; stay <= 64 KiB so that two workgroups share a CU
    int fr = (64 * 1024) / (2 * M * 8);
    if (fr < 1) fr = 1;
    if (fr > 16) fr = 16;
    return fr;
}

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_stft(const float* x, float* out, const float* window, const float* tw_half, const float* tw_full,
                         int n_sig, int L, int n_fft, int hop, int n_frames, int n_bins_out, int64_t o_sig,
                         int64_t o_im, int64_t o_bin, int64_t o_frame, void* stream) {
    if (!x || !out || !window || !tw_half || !tw_full) return fail(AICG_E_ARG, "aicg_stft: null pointer");
    if (n_fft < 4 || (n_fft & 1) || n_fft > 16384 || !smooth235(n_fft / 2))
        return fail(AICG_E_SHAPE, "aicg_stft: n_fft=%d must be even, <=16384 and n_fft/2 = 2^a 3^b 5^c", n_fft);
    if (n_sig < 0 || n_frames < 0 || hop < 1 || n_bins_out < 1 || n_bins_out > n_fft / 2 + 1 || L <= n_fft / 2)
        return fail(AICG_E_SHAPE, "aicg_stft: bad shape (L=%d must exceed n_fft/2 for reflect padding)", L);
    if ((long)(n_frames - 1) * hop > (long)L) return fail(AICG_E_SHAPE, "aicg_stft: n_frames too large for L");
    if (n_sig == 0 || n_frames == 0) return AICG_OK;
    const int M = n_fft / 2;
    StftArgs p{x, out, window, (const float2*)tw_half, (const float2*)tw_full, n_sig, L, n_fft, hop, n_frames,
               n_bins_out, pick_frames_per_block(M), (long)o_sig, (long)o_im, (long)o_bin, (long)o_frame};
    if (((uintptr_t)window & 7) == 0) {   // the plan kernels read the window as float2
        const int rc = with_plan(M, p, (hipStream_t)stream, [&](auto plan) { return launch_stft_reg<decltype(plan)>(p, (hipStream_t)stream); });
        if (rc <= 0) return rc;
    }
    const size_t lds = (size_t)2 * p.fr_per_block * M * sizeof(float2);
    const long total = (long)n_sig * n_frames;
    const unsigned grid = (unsigned)((total + p.fr_per_block - 1) / p.fr_per_block);
    if (lds > 64 * 1024)
        allow_dynamic_lds((const void*)stft_kernel, lds);
    hipLaunchKernelGGL(stft_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("stft_kernel");
}

extern "C" int aicg_istft_frames(const float* spec, float* frames, const float* window, const float* tw_half,
                                 const float* tw_full, int n_sig, int n_fft, int n_frames, int n_bins_in,
                                 int64_t i_sig, int64_t i_im, int64_t i_bin, int64_t i_frame, void* stream) {
    if (!spec || !frames || !window || !tw_half || !tw_full) return fail(AICG_E_ARG, "aicg_istft_frames: null pointer");
    if (n_fft < 4 || (n_fft & 1) || n_fft > 16384 || !smooth235(n_fft / 2))
        return fail(AICG_E_SHAPE, "aicg_istft_frames: unsupported n_fft=%d", n_fft);
    if (n_bins_in < 1 || n_bins_in > n_fft / 2 + 1 || n_sig < 0 || n_frames < 0)
        return fail(AICG_E_SHAPE, "aicg_istft_frames: bad shape");
    if (n_sig == 0 || n_frames == 0) return AICG_OK;
    const int M = n_fft / 2;
    IstftArgs p{spec, frames, window, (const float2*)tw_half, (const float2*)tw_full, n_sig, n_fft, n_frames,
                n_bins_in, pick_frames_per_block(M), (long)i_sig, (long)i_im, (long)i_bin, (long)i_frame};
    if (((uintptr_t)window & 7) == 0 && ((uintptr_t)frames & 7) == 0) {
        const int rc = with_plan(M, p, (hipStream_t)stream, [&](auto plan) { return launch_istft_reg<decltype(plan)>(p, (hipStream_t)stream); });
        if (rc <= 0) return rc;
    }
    const size_t lds = (size_t)2 * p.fr_per_block * M * sizeof(float2);
    const long total = (long)n_sig * n_frames;
    const unsigned grid = (unsigned)((total + p.fr_per_block - 1) / p.fr_per_block);
    if (lds > 64 * 1024)
        allow_dynamic_lds((const void*)istft_frames_kernel, lds);
    hipLaunchKernelGGL(istft_frames_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    return check_launch("istft_frames_kernel");
}

extern "C" int aicg_istft_ola(const float* frames, const float* window, float* out, int n_sig, int L, int n_fft,
                              int hop, int n_frames, void* stream) {
    if (!frames || !window || !out) return fail(AICG_E_ARG, "aicg_istft_ola: null pointer");
    if (n_sig < 0 || L < 0 || hop < 1 || n_frames < 1) return fail(AICG_E_SHAPE, "aicg_istft_ola: bad shape");
    if ((long)L + n_fft / 2 > (long)(n_frames - 1) * hop + n_fft)
        return fail(AICG_E_SHAPE, "aicg_istft_ola: L=%d not covered by %d frames", L, n_frames);
    const long total = (long)n_sig * L;
    if (total == 0) return AICG_OK;
    if ((L & 3) == 0 && (hop & 3) == 0 && (n_fft & 7) == 0 && ((uintptr_t)frames & 15) == 0 && ((uintptr_t)window & 15) == 0 &&
        ((uintptr_t)out & 15) == 0) {
        const unsigned grid4 = (unsigned)lmin((total / 4 + 255) / 256, 256L * 16);
        hipLaunchKernelGGL(istft_ola4_kernel, dim3(grid4), dim3(256), 0, (hipStream_t)stream, frames, window, out, n_sig, L, n_fft, hop,
                           n_frames);
        return check_launch("istft_ola4_kernel");
    }
    unsigned grid = (unsigned)lmin((total + 255) / 256, 256L * 8);
    hipLaunchKernelGGL(istft_ola_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, frames, window, out, n_sig, L,
                       n_fft, hop, n_frames);
    return check_launch("istft_ola_kernel");
}
