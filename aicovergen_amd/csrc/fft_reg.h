// Register-resident mixed-radix FFT passes with COMPILE-TIME plans, for the framed STFT / iSTFT (stft.hip).
//
// A frame of M complex points (M = n_fft / 2: two real samples packed per complex point) is transformed in three Stockham passes
// of radix R0, R1, R2 (M = R0 R1 R2, each radix <= 32): a thread holds one radix-R butterfly in registers, so a pass is
//   LDS (or, for the first pass, HBM) -> R registers -> twiddles -> DFT_R on constants -> LDS (or, for the last pass, HBM)
// instead of the log2-deep walk of radix-4 / -2 / -3 / -5 passes over two ping-pong LDS buffers the run-time-plan kernel takes
// (six passes for M = 3840, each with integer divisions per butterfly and a branch on the radix).  Everything that depends on the
// plan is a template constant: strides, butterfly counts, the small-DFT roots (constexpr tables: immediates after unrolling).
// The only run-time table is the M-entry twiddle table the caller already has (fp64-evaluated, rounded once).
//
// LDS layout: element e of a frame lives at slot e + (e >> 4).  The first pass writes R0 consecutive elements per lane (lane j ->
// elements j R0 ... j R0 + R0 - 1): unpadded, a 16-lane ds_write_b64 group would hit one bank pair 16 times; padded, lane stride
// is 2 (R0 + R0 / 16) dwords and the 16 lanes of a group cover distinct banks.  All other accesses are lane-contiguous.
#pragma once
#include "common.h"

namespace aicg {
namespace fftc {

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double csin(double x) { double t = x, s = x; for (int i = 1; i < 18; ++i) { t *= -x * x / ((2.0 * i) * (2.0 * i + 1.0)); s += t; } return s; }
constexpr double ccos(double x) { double t = 1.0, s = 1.0; for (int i = 1; i < 18; ++i) { t *= -x * x / ((2.0 * i - 1.0) * (2.0 * i)); s += t; } return s; }

// exp(-2 pi i k / N) = c[k] + i s[k], evaluated in double at compile time (argument reduced to (-pi, pi])
template <int N>
struct Roots {
    float c[N], s[N];
    constexpr Roots() : c{}, s{} {
        for (int k = 0; k < N; ++k) {
            double a = 2.0 * kPi * k / N;
            if (a > kPi) a -= 2.0 * kPi;
            c[k] = (float)ccos(a);
            s[k] = (float)(-csin(a));
        }
    }
};
template <int N>
struct RootsOf { static constexpr Roots<N> v{}; };

__device__ __forceinline__ float2 fadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 fsub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 fmulc(float2 a, float c, float s) { return make_float2(a.x * c - a.y * s, a.x * s + a.y * c); }
// multiplication by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 rot(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

// In-register DFT of R points, forward (exp(-2 pi i nk / R)) or inverse (conjugate roots), natural order in and out.
template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<1, INV> { static __device__ __forceinline__ void run(float2 (&)[1]) {} };

template <bool INV>
struct Dft<2, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[2]) {
        const float2 a = v[0], b = v[1];
        v[0] = fadd(a, b);
        v[1] = fsub(a, b);
    }
};

template <bool INV>
struct Dft<3, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[3]) {
        const float2 t1 = fadd(v[1], v[2]);
        const float2 t2 = make_float2(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
        const float2 d = fsub(v[1], v[2]);
        const float2 t3 = rot<INV>(make_float2(0.86602540378443864676f * d.x, 0.86602540378443864676f * d.y));
        v[0] = fadd(v[0], t1);
        v[1] = fadd(t2, t3);
        v[2] = fsub(t2, t3);
    }
};

template <bool INV>
struct Dft<4, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[4]) {
        const float2 a = fadd(v[0], v[2]), b = fsub(v[0], v[2]), c = fadd(v[1], v[3]), d = rot<INV>(fsub(v[1], v[3]));
        v[0] = fadd(a, c);
        v[1] = fadd(b, d);
        v[2] = fsub(a, c);
        v[3] = fsub(b, d);
    }
};

template <bool INV>
struct Dft<5, INV> {
    static __device__ __forceinline__ void run(float2 (&v)[5]) {
        constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const float2 a1 = fadd(v[1], v[4]), a2 = fadd(v[2], v[3]), b1 = fsub(v[1], v[4]), b2 = fsub(v[2], v[3]);
        const float2 r1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
        const float2 r2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
        const float2 i1 = rot<INV>(make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
        const float2 i2 = rot<INV>(make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
        v[0] = fadd(v[0], fadd(a1, a2));
        v[1] = fadd(r1, i1);
        v[4] = fsub(r1, i1);
        v[2] = fadd(r2, i2);
        v[3] = fsub(r2, i2);
    }
};

constexpr int inner_radix(int r) { return (r % 4 == 0 && r > 4) ? 4 : (r % 5 == 0 && r > 5) ? 5 : (r % 3 == 0 && r > 3) ? 3 : 2; }

// Composite radix R = A B (Cooley-Tukey inside the registers): n = A n2 + n1, k = B k1 + k2:
//   X[B k1 + k2] = sum_n1 W_A^{n1 k1} ( W_R^{n1 k2} sum_n2 x[A n2 + n1] W_B^{n2 k2} )
template <int R, bool INV>
struct Dft {
    static constexpr int B = inner_radix(R), A = R / B;
    static_assert(A * B == R && A > 1, "radix must factor into 2, 3, 4, 5");
    static __device__ __forceinline__ void run(float2 (&v)[R]) {
        float2 y[A][B];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) {
            float2 t[B];
#pragma unroll
            for (int n2 = 0; n2 < B; ++n2) t[n2] = v[A * n2 + n1];
            Dft<B, INV>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) {
                const int e = (n1 * k2) % R;
                y[n1][k2] = e == 0 ? t[k2] : fmulc(t[k2], RootsOf<R>::v.c[e], INV ? -RootsOf<R>::v.s[e] : RootsOf<R>::v.s[e]);
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) {
            float2 t[A];
#pragma unroll
            for (int n1 = 0; n1 < A; ++n1) t[n1] = y[n1][k2];
            Dft<A, INV>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < A; ++k1) v[B * k1 + k2] = t[k1];
        }
    }
};

// LDS slot of frame element e
__device__ __forceinline__ int slot(int e) { return e + (e >> 4); }
constexpr int frame_slots(int m) { return m + (m >> 4) + 1; }

template <int R0_, int R1_, int R2_, int FR_>
struct Plan {
    static constexpr int R0 = R0_, R1 = R1_, R2 = R2_, FR = FR_;
    static constexpr int M = R0 * R1 * R2;
    static constexpr int T0 = M / R0, T1 = M / R1, T2 = M / R2;           // butterflies per pass
    static constexpr int TMAX = T0 > T1 ? (T0 > T2 ? T0 : T2) : (T1 > T2 ? T1 : T2);
    static constexpr int TPF = (TMAX + 63) / 64 * 64;                     // threads per frame
    static constexpr int NT = TPF * FR;                                   // threads per workgroup
    static constexpr int SLOTS = frame_slots(M);
    static_assert(NT <= 1024 && TPF >= TMAX, "plan does not fit a workgroup");
};

// One pass after the first: butterfly j of the frame in `buf` (Ns = product of the radices already applied), result left in v.
template <int R, int M, int NS, bool INV>
__device__ __forceinline__ void load_pass(const float2* __restrict__ buf, const float2* __restrict__ tw, int j, float2 (&v)[R]) {
    constexpr int T = M / R, TSTEP = M / (NS * R);
    const int k = j % NS;   // NS is a compile-time constant (a power of two in every instantiated plan): a mask
#pragma unroll
    for (int t = 0; t < R; ++t) v[t] = buf[slot(j + t * T)];
    if (k != 0) {
#pragma unroll
        for (int t = 1; t < R; ++t) {
            float2 w = tw[t * k * TSTEP];
            if (INV) w.y = -w.y;
            v[t] = fmulc(v[t], w.x, w.y);
        }
    }
    Dft<R, INV>::run(v);
}

// Stockham output position of result q of butterfly j
template <int R, int NS>
__device__ __forceinline__ int out_pos(int j, int q) {
    const int k = j % NS;
    return (j - k) * R + k + q * NS;
}

}  // namespace fftc
}  // namespace aicg
