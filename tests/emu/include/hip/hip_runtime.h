// TEST INFRASTRUCTURE ONLY -- never part of the product.
//
// A stand-in for <hip/hip_runtime.h> that lets the *unmodified* gfx950 kernel sources in
// aicovergen_amd/csrc/ be compiled by the host clang++ and executed on CPU cores, so that kernel
// index arithmetic, LDS staging, barriers and MFMA fragment layouts can be checked in the
// GPU-less build container (pytest -m "not gpu").  The product library (libaicg_hip.so) is
// always built by hipcc against the real HIP headers; nothing under tests/emu/ is linked into it,
// and the Python package refuses to run without the real library (aicovergen_amd/_lib.py).
//
// Execution model: every workgroup runs as a set of cooperatively scheduled fibers (one fiber per
// work-item) on one OS thread; __syncthreads() and the wave-collective operations (shuffles, MFMA)
// are rendezvous points.  A wave is 64 consecutive work-items, as on CDNA4.  MFMA fragment layouts
// follow /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <utility>

#define AICG_EMULATED 1

// ---- qualifiers -------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __constant__ static
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(::emu::dyn_smem());

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipPeekAtLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8

// ---- vector types ------------------------------------------------------------------------------
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline int2 make_int2(int x, int y) { return {x, y}; }
inline double2 make_double2(double x, double y) { return {x, y}; }

namespace emu {
struct Fiber;
struct Ids { dim3 tid; };
extern thread_local Fiber* cur;
extern thread_local dim3 tl_block_idx;
extern dim3 g_grid, g_block;
dim3& cur_tid();
void* dyn_smem();
void block_barrier();
void wave_barrier();
int lane_id();
// exchange buffers of the current wave (64 slots of 16 bytes)
void* wave_slot(int lane);
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void yield_any();  // let another work-item of the workgroup run (used by spin loops)
}  // namespace emu

#define threadIdx (::emu::cur_tid())
#define blockIdx (::emu::tl_block_idx)
#define blockDim (::emu::g_block)
#define gridDim (::emu::g_grid)
#define warpSize 64

inline void __syncthreads() { ::emu::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <typename T>
inline T emu_wave_read(T v, int src_lane) {
    static_assert(sizeof(T) <= 16, "wave exchange slot is 16 bytes");
    const int lane = ::emu::lane_id();
    memcpy(::emu::wave_slot(lane), &v, sizeof(T));
    ::emu::wave_barrier();
    T out;
    memcpy(&out, ::emu::wave_slot(src_lane & 63), sizeof(T));
    ::emu::wave_barrier();
    return out;
}
template <typename T> inline T __shfl(T v, int src, int width = 64) {
    const int lane = ::emu::lane_id();
    const int base = lane & ~(width - 1);
    return emu_wave_read(v, base + (src & (width - 1)));
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
    const int lane = ::emu::lane_id();
    int src = lane ^ mask;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return emu_wave_read(v, src);
}
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int lane = ::emu::lane_id();
    int src = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return emu_wave_read(v, src);
}
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int lane = ::emu::lane_id();
    int src = lane - (int)d;
    if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return emu_wave_read(v, src);
}
inline unsigned long long __ballot(int pred) {
    unsigned long long mine = pred ? 1ull : 0ull;
    unsigned long long m = 0;
    // gather through 64 reads (slow, fine for tests)
    const int lane = ::emu::lane_id();
    memcpy(::emu::wave_slot(lane), &mine, sizeof(mine));
    ::emu::wave_barrier();
    for (int l = 0; l < 64; ++l) {
        unsigned long long b;
        memcpy(&b, ::emu::wave_slot(l), sizeof(b));
        m |= (b & 1ull) << l;
    }
    ::emu::wave_barrier();
    return m;
}

// ---- MFMA (f32 in / f32 accumulate).  Layouts: cdna_hip_programming.md section 3 -----------------
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  k-ordered fmaf chain.
inline emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    const int lane = ::emu::lane_id();
    float ab[2] = {a, b};
    memcpy(::emu::wave_slot(lane), ab, sizeof(ab));
    ::emu::wave_barrier();
    emu_f32x16 d = c;
    const int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av[2], bv[2];
            memcpy(av, ::emu::wave_slot(k * 32 + row), sizeof(av));
            memcpy(bv, ::emu::wave_slot(k * 32 + col), sizeof(bv));
            acc = fmaf(av[0], bv[1], acc);
        }
        d[r] = acc;
    }
    ::emu::wave_barrier();
    return d;
}
// v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15]; D: col = lane&15, row = (lane>>4)*4 + reg
inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    const int lane = ::emu::lane_id();
    float ab[2] = {a, b};
    memcpy(::emu::wave_slot(lane), ab, sizeof(ab));
    ::emu::wave_barrier();
    emu_f32x4 d = c;
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av[2], bv[2];
            memcpy(av, ::emu::wave_slot(k * 16 + row), sizeof(av));
            memcpy(bv, ::emu::wave_slot(k * 16 + col), sizeof(bv));
            acc = fmaf(av[0], bv[1], acc);
        }
        d[r] = acc;
    }
    ::emu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_s_sleep(x) ::emu::yield_any()
// agent-scope atomics on global memory: workgroups run on different OS threads in the emulator
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)   /* callers pass wave-uniform values */
#define __builtin_amdgcn_wave_barrier() ::emu::wave_barrier()   /* lanes of a wave run in lockstep on the hardware: rendezvous here */
#define __builtin_amdgcn_s_barrier() ::emu::block_barrier()

// ---- device math ---------------------------------------------------------------------------------
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline double rsqrt(double x) { return 1.0 / sqrt(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline void sincospif(float x, float* s, float* c) { *s = sinf((float)M_PI * x); *c = cosf((float)M_PI * x); }
inline float sinpif(float x) { return (float)sin(M_PI * (double)x); }
inline float cospif(float x) { return (float)cos(M_PI * (double)x); }
inline double sinpi(double x) { return sin(M_PI * x); }
inline double cospi(double x) { return cos(M_PI * x); }

inline float atomicAdd(float* p, float v) {
    float old, nw;
    do { old = *p; nw = old + v; } while (!__atomic_compare_exchange(p, &old, &nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// ---- kernel launch -------------------------------------------------------------------------------
template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t,
                               Args&&... args) {
    std::tuple<KArgs...> targs(std::forward<Args>(args)...);
    ::emu::launch(grid, block, shmem, [&]() { std::apply(kernel, targs); });
}
