// Shared helpers for the gfx950 kernels of libaicg_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/aicg.h"

namespace aicg {

// error plumbing: the C ABI returns negative codes and keeps a per-thread message for aicg_last_error()
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AICG_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return AICG_OK;
}

// Raise a kernel's dynamic-LDS limit above the 64 KB default, once per kernel (hipFuncSetAttribute is a blocking runtime
// call: issued per launch it serialises streams -- it stalled pipeline()'s f0 side stream against the main stream).
void allow_dynamic_lds(const void* kernel, size_t bytes);

__host__ __device__ inline int idiv_up(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline long ldiv_up(long a, long b) { return (a + b - 1) / b; }
__host__ __device__ inline int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ inline long lmin(long a, long b) { return a < b ? a : b; }
__host__ __device__ inline long lmax(long a, long b) { return a > b ? a : b; }

// activations shared by conv epilogues and elementwise kernels (codes: include/aicg.h).
// The transcendental ones are kept out of line: the conv kernels instantiate the activation ~80 times per
// template variant and inlining erff/tanhf there multiplies the code size past the instruction cache.
__device__ __attribute__((noinline)) inline float apply_act_slow(float v, int act, float slope) {
    switch (act) {
        case AICG_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        case AICG_ACT_TANH: return tanhf(v);
        case AICG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case AICG_ACT_LOGCLAMP: return logf(fmaxf(v, slope));
        default: return v;
    }
}
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == AICG_ACT_NONE) return v;
    if (act == AICG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == AICG_ACT_LRELU) return v > 0.f ? v : v * slope;
    return apply_act_slow(v, act, slope);
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the global loads a wave still
// has in flight into registers (the producer waves of conv_ws_kernel keep one stage of loads outstanding across it).
__device__ __forceinline__ void lds_barrier() {
#ifdef AICG_EMULATED
    __builtin_amdgcn_s_barrier();
#else
    // The wait goes through the builtin, not the asm string: the compiler's counter model then knows that no scalar load is
    // pending after a barrier.  With an opaque wait a kernel-argument s_load hoisted above the K loop stayed "pending" for the
    // whole loop in that model, and with two event types on lgkmcnt every fragment wait in the MFMA loop degraded to
    // lgkmcnt(0) -- the reads of k-step s+1, issued just before, were drained ahead of the MFMAs of step s (r2 ISA audit).
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); vmcnt / expcnt untouched (gfx9 encoding: [11:8] lgkm, [3:0]+[15:14] vm, [6:4] exp)
    asm volatile("s_barrier" ::: "memory");
#endif
}

// block->XCD aware remap (guide T1, bijective form): consecutive logical ids share an XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned nx = 8;
    if (nwg < nx * 2) return bid;
    const unsigned q = nwg / nx, r = nwg % nx;
    const unsigned xcd = bid % nx, pos = bid / nx;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + pos;
}

}  // namespace aicg
