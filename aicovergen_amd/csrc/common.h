// Shared helpers for the gfx950 kernels of libaicg_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/aicg.h"

// A/B switches.  Development builds (-DAICG_DEV_SWITCHES: the CPU emulator of tests/emu, tools/'s private library) read them from
// the environment once; in the product library they are compile-time constants -- the dispatcher carries no getenv and the kernel
// generations only a non-default switch can select are not linked.
#ifdef AICG_DEV_SWITCHES
#include <cstdlib>
#define AICG_SWITCH(var, name, dflt) static const long var = getenv(name) ? atol(getenv(name)) : (long)(dflt)
#else
#define AICG_SWITCH(var, name, dflt) constexpr long var = (long)(dflt)
#endif

// waves per SIMD a kernel's register allocation is held to (exactly n: hipcc neither takes more registers nor spills for a higher count)
#ifdef AICG_EMULATED
#define AICG_WAVES_PER_SIMD(n)
#else
#define AICG_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

namespace aicg {

// error plumbing: the C ABI returns negative codes and keeps a per-thread message for aicg_last_error()
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

// name of the kernel family the calling thread launched last (aicg_last_launch(): tests assert WHICH kernel a shape was routed to)
void note_launch(const char* what);

inline int check_launch(const char* what) {
    note_launch(what);
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
// erf(a) without a branch, < 1 ulp against the exact function over the whole line (checked at 3 million points in
// tests/test_kernels_misc.py): two minimax fits in the published single-precision form -- |a| <= 0.921875: a + a q(a^2); above:
// 1 - exp(-(t + t r(t))), t = |a| -- both evaluated (17 FMAs + one v_exp_f32), one select.  The library erff costs ~3x as much and
// branches per lane; GELU sits in the epilogue of HuBERT's fc1 GEMM and of its feature-extractor convolutions, 40 M evaluations per call.
__device__ __forceinline__ float fast_erff(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(0x1.222900p-16f, t, -0x1.91d2ccp-12f);
    const float u = fmaf(0x1.fd1336p-09f, t, -0x1.8d6300p-06f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, 0x1.b55cb0p-4f);
    r = fmaf(r, t, 0x1.450aa0p-1f);
    r = fmaf(r, t, 0x1.079d0cp-3f);
    r = fmaf(r, t, t);
#ifdef AICG_EMULATED
    r = 1.f - expf(-r);
#else
    r = 1.f - __builtin_amdgcn_exp2f(r * -1.44269504088896340736f);   // (r in [0.9, inf): v_exp_f32 needs no range handling here)
#endif
    r = copysignf(r, a);
    float q = -0x1.3a1a82p-11f;
    q = fmaf(q, s, 0x1.473f48p-08f);
    q = fmaf(q, s, -0x1.b68bd2p-06f);
    q = fmaf(q, s, 0x1.ce1a46p-04f);
    q = fmaf(q, s, -0x1.8126e0p-02f);
    q = fmaf(q, s, 0x1.06eba6p-03f);
    q = fmaf(q, a, a);
    return t > 0.921875f ? r : q;
}
// exact-erf GELU (fairseq / HF HuBERT "gelu")
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + fast_erff(v * 0.70710678118654752440f)); }

__device__ __attribute__((noinline)) inline float apply_act_slow(float v, int act, float slope) {
    switch (act) {
        case AICG_ACT_GELU: return gelu_erf(v);
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

// Buffer-resource loads: SGPR base + one 32-bit VGPR byte offset per load (a 64-bit global_load address costs two VGPRs and
// a carry chain per load), and the hardware range check returns 0 for offsets >= num_records -- zero padding and absent
// channels need no mask, no select and no exec juggling: invalid slots simply carry the offset kBufOob.
#ifdef AICG_EMULATED
struct BufRsrc { const char* base; unsigned num_records; };
__device__ __forceinline__ BufRsrc make_buf(const void* base, unsigned num_bytes) { return BufRsrc{(const char*)base, num_bytes}; }
__device__ __forceinline__ float buf_load_f32(const BufRsrc& r, unsigned voff) {
    if ((unsigned long)voff + 4 > r.num_records) return 0.f;
    return *reinterpret_cast<const float*>(r.base + voff);
}
__device__ __forceinline__ float4 buf_load_f32x4(const BufRsrc& r, unsigned voff) {
    if ((unsigned long)voff + 16 > r.num_records) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(r.base + voff);
}
__device__ __forceinline__ float buf_load_f32_s(const BufRsrc& r, unsigned voff, unsigned soff) {
    if ((unsigned long)voff + 4 > r.num_records) return 0.f;   // like the hardware: the scalar offset is not range checked
    return *reinterpret_cast<const float*>(r.base + voff + soff);
}
__device__ __forceinline__ float4 buf_load_f32x4_s(const BufRsrc& r, unsigned voff, unsigned soff) {   // dword-aligned
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned long)voff + 16 <= r.num_records) __builtin_memcpy(&v, r.base + voff + soff, 16);
    return v;
}
#else
// The raw-buffer intrinsics are bound by name (the clang builtin __builtin_amdgcn_raw_buffer_load_b128 of ROCm 7.2 lowers to
// a 32-bit load and splats it); the resource is the classic 4-dword descriptor {base[47:0], stride 0, num_records, flags}.
typedef int buf_i32x4 __attribute__((ext_vector_type(4)));
typedef float buf_f32x4 __attribute__((ext_vector_type(4)));
__device__ float llvm_amdgcn_raw_buffer_load_f32(buf_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ buf_f32x4 llvm_amdgcn_raw_buffer_load_v4f32(buf_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
struct BufRsrc { buf_i32x4 d; };
__device__ __forceinline__ BufRsrc make_buf(const void* base, unsigned num_bytes) {
    const unsigned long a = (unsigned long)base;
    BufRsrc r;
    r.d.x = (int)(unsigned)(a & 0xffffffffu);
    r.d.y = (int)(unsigned)((a >> 32) & 0xffffu);  // stride 0, no swizzle: raw buffer
    r.d.z = (int)num_bytes;
    r.d.w = 0x00020000;                            // 32-bit data format, range checking on raw offsets
    return r;
}
__device__ __forceinline__ float buf_load_f32(const BufRsrc& r, unsigned voff) { return llvm_amdgcn_raw_buffer_load_f32(r.d, (int)voff, 0, 0); }
// voff + a wave-uniform byte offset in an SGPR: no VALU add per load.  The hardware range-checks voff alone (the scalar offset is
// excluded), so the caller guarantees that base + voff + soff is inside the tensor whenever voff is in range.
__device__ __forceinline__ float buf_load_f32_s(const BufRsrc& r, unsigned voff, unsigned soff) {
    return llvm_amdgcn_raw_buffer_load_f32(r.d, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float4 buf_load_f32x4_s(const BufRsrc& r, unsigned voff, unsigned soff) {   // dword-aligned address
    const buf_f32x4 v = llvm_amdgcn_raw_buffer_load_v4f32(r.d, (int)voff, (int)soff, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 buf_load_f32x4(const BufRsrc& r, unsigned voff) {
    const buf_f32x4 v = llvm_amdgcn_raw_buffer_load_v4f32(r.d, (int)voff, 0, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}
#endif
static constexpr unsigned kBufOob = 0x80000000u;  // any byte offset >= 2^31 is out of range for the buffers made here

// ---- split-precision (bf16 hi + bf16 lo) helpers -----------------------------------------------------------------------------
// x ~= hi + lo with hi = bf16(x) (round to nearest even) and lo = bf16(x - hi): 16 significand bits of x survive; products are
// formed as hi*hi + hi*lo + lo*hi on the bf16 matrix pipe with fp32 accumulation (the lo*lo term, <= 2^-16 relative, is dropped).
__device__ __forceinline__ unsigned bf16_rne_bits(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;   // (finite inputs; NaN payloads are not preserved)
}
__device__ __forceinline__ void split_bf16(float x, unsigned& hi, unsigned& lo) {
    hi = bf16_rne_bits(x);
    lo = bf16_rne_bits(x - __builtin_bit_cast(float, hi << 16));
}
// two values at once: word = bf16(v0) | bf16(v1) << 16 for the hi and the lo parts (v_cvt_pk_bf16_f32 on gfx950: 5 VALU ops a pair)
__device__ __forceinline__ void split_bf16_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
#ifdef AICG_EMULATED
    unsigned h0, l0, h1, l1;
    split_bf16(v0, h0, l0);
    split_bf16(v1, h1, l1);
    hi = h0 | (h1 << 16);
    lo = l0 | (l1 << 16);
#else
    // five VALU ops a pair (the (__bf16) casts compiled to a second, redundant conversion for the hi << 16 term)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    const float d0 = v0 - __builtin_bit_cast(float, hi << 16), d1 = v1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(d0), "v"(d1));
#endif
}
// 8 bf16 (k = 8 half .. 8 half + 7 of a 16-row K group) packed in a float4: element e in bits 16 (e & 1) of word e >> 1
typedef float mfma_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ mfma_f32x16 mfma_bf16_32x32x16(const float4& a, const float4& b, mfma_f32x16 c) {
#ifdef AICG_EMULATED
    const unsigned aw[4] = {__builtin_bit_cast(unsigned, a.x), __builtin_bit_cast(unsigned, a.y), __builtin_bit_cast(unsigned, a.z),
                            __builtin_bit_cast(unsigned, a.w)};
    const unsigned bw[4] = {__builtin_bit_cast(unsigned, b.x), __builtin_bit_cast(unsigned, b.y), __builtin_bit_cast(unsigned, b.z),
                            __builtin_bit_cast(unsigned, b.w)};
    for (int e = 0; e < 8; ++e) {   // 8 two-row contractions of the fp32 MFMA model: lane half h supplies k = 8 h + e
        const float av = __builtin_bit_cast(float, (aw[e >> 1] >> (16 * (e & 1))) << 16);
        const float bv = __builtin_bit_cast(float, (bw[e >> 1] >> (16 * (e & 1))) << 16);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
    }
    return c;
#else
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// ---- opt-in fp16 matrix arithmetic (aicg_conv_desc.split == 2: the reference's is_half mode on the RVC half, src/rvc.py:103-104,137-138):
// operands rounded to fp16 (round to nearest even, v_cvt_pk_f16_f32) in registers right in front of the MFMA, fp32 accumulation, fp32
// activations in HBM.  Four fp16 (k = 4 half .. 4 half + 3 of an 8-row K group) packed in two dwords: element e in bits 16 (e & 1) of
// word e >> 1.
struct H4 { unsigned x, y; };
__device__ __forceinline__ H4 pack_f16x4(float v0, float v1, float v2, float v3) {
    H4 r;
#ifdef AICG_EMULATED
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t a = {v0, v1}, b = {v2, v3};
    r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(a, h2_t));
    r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(b, h2_t));
#else
    // (inline asm: through __builtin_convertvector hipcc kept the fragment quads the values come from in scratch memory.  The hazard
    //  recogniser does not look inside: a VALU result needs two wait states before an MFMA reads it, hence the s_nop)
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\ts_nop 1" : "=&v"(r.x), "=&v"(r.y) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
#endif
    return r;
}
__device__ __forceinline__ mfma_f32x16 mfma_f16_32x32x8(const H4& a, const H4& b, mfma_f32x16 c) {
#ifdef AICG_EMULATED
    const unsigned aw[2] = {a.x, a.y}, bw[2] = {b.x, b.y};
    for (int e = 0; e < 4; ++e) {   // four two-row contractions of the fp32 MFMA model: lane half h supplies k = 4 h + e
        const unsigned short ab = (unsigned short)(aw[e >> 1] >> (16 * (e & 1))), bb = (unsigned short)(bw[e >> 1] >> (16 * (e & 1)));
        const float av = (float)__builtin_bit_cast(_Float16, ab), bv = (float)__builtin_bit_cast(_Float16, bb);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
    }
    return c;
#else
    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(f16x4_t, a), __builtin_bit_cast(f16x4_t, b), c, 0, 0, 0);
#endif
}

// value of the lane whose id differs in bit 0 / bit 1 (exchange inside a quad of lanes): one DPP move on the hardware
__device__ __forceinline__ float quad_xor1(float v) {
#ifdef AICG_EMULATED
    return __shfl_xor(v, 1, 64);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
#endif
}
__device__ __forceinline__ float quad_xor2(float v) {
#ifdef AICG_EMULATED
    return __shfl_xor(v, 2, 64);
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true));
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
