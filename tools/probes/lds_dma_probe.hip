// What does the CU's global -> LDS path sustain, and what does a wave pay to issue into it?  conv_w2d stages 47 KiB per 6 144 MFMA
// cycles and its in-kernel timeline shows a ~3 000-cycle burst in which no wave computes (profiles/NOTES.md 2.9).  This probe
// separates the candidates: one workgroup of eight waves (two per SIMD, the kernel's shape) per CU, every stage = 48 pieces of
// 1 KiB (64 lanes x 16 B), 96 v_mfma_f32_16x16x4_f32 per wave and stage.
//   source : L2  -- the same 48 KiB for every workgroup and stage (the kernel's weights);  HBM -- fresh bytes every stage
//   path   : buffer_load_dwordx4 ... lds | global_load_lds_dwordx4 | global_load_dwordx4 into registers + ds_write_b128
//   issue  : one burst at the top of the stage (all eight waves at once) | one piece per 16 MFMAs, same position in every wave |
//            one piece per 16 MFMAs, position staggered by wave | only the upper four waves issue (12 pieces each)
// Reported per variant: cycles per stage (s_memtime of wave 0, averaged over the stages), bytes per clock and CU, and the stage's
// MFMA floor (96 x 2 waves x 32 cycles = 6 144).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/lds_dma_probe tools/probes/lds_dma_probe.hip && tools/probes/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

enum { SRC_L2 = 0, SRC_HBM = 1 };
enum { PATH_BUF = 0, PATH_GLDS = 1, PATH_REG = 2, PATH_NONE = 3 };
enum { ISS_BURST = 0, ISS_SPREAD = 1, ISS_STAGGER = 2, ISS_UPPER4 = 3 };

__device__ __forceinline__ void dma_buf(i32x4 rs, unsigned voff, unsigned soff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_glds(const float* sbase, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}

template <int SRC, int PATH, int ISS, int MFMAS>
__global__ void __launch_bounds__(512) probe(const float* src, float* out, int stages) {
    extern __shared__ float4 smem4[];
    float* smem = reinterpret_cast<float*>(smem4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PIECES = 48, STAGE_F = PIECES * 256;           // floats per stage buffer (48 KiB)
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1e-9f * (float)(lane & 7), b = 0.25f;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned voff = 16u * (unsigned)lane;
    // pieces of this wave: ISS_UPPER4: waves 4..7 take 12 each, the others none; otherwise 6 each
    const int npw = ISS == ISS_UPPER4 ? (wave >= 4 ? 12 : 0) : 6;
    const int p0 = ISS == ISS_UPPER4 ? (wave - 4) * 12 : wave * 6;
    long long t0 = 0;
    float4 regs[6];
    for (int st = 0; st < stages + 1; ++st) {
        if (st == 1 && tid == 0) t0 = clock64();
        const int bufi = st % 3;
        const long sbase_f = SRC == SRC_L2 ? 0L : ((long)blockIdx.x * (stages + 1) + st) * STAGE_F;
        const float* sb = src + sbase_f;
        i32x4 rs;
        rs.x = (int)(uintptr_t)sb;
        rs.y = (int)(((uintptr_t)sb >> 32) & 0xffff);
        rs.z = STAGE_F * 4;
        rs.w = 0x00020000;
        rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y);
        const unsigned ldsb = lds0 + (unsigned)bufi * STAGE_F * 4;
        auto piece = [&](int e) __attribute__((always_inline)) {
            if (e >= npw) return;
            const unsigned pc = (unsigned)(p0 + e);
            if constexpr (PATH == PATH_BUF) dma_buf(rs, voff, 1024u * pc, ldsb + 1024u * pc);
            else if constexpr (PATH == PATH_GLDS) dma_glds(sb + 256 * pc, voff, ldsb + 1024u * pc);
            else if constexpr (PATH == PATH_REG) regs[e % 6] = *reinterpret_cast<const float4*>(sb + 256 * pc + 4 * lane);
        };
        auto commit = [&](int e) __attribute__((always_inline)) {
            if constexpr (PATH == PATH_REG) {
                if (e < npw) *reinterpret_cast<float4*>(smem + bufi * STAGE_F + 256 * (p0 + e) + 4 * lane) = regs[e % 6];
            }
        };
        if constexpr (ISS == ISS_BURST || ISS == ISS_UPPER4) {
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                piece(e);
                if constexpr (PATH == PATH_REG) { if (e % 6 == 5) { for (int q = e - 5; q <= e; ++q) commit(q); } }
            }
        }
        // the stage's MFMAs in blocks of 16; spread forms put one piece in front of block k (staggered: wave w shifts by w mod 4 MFMAs)
#pragma unroll
        for (int blk = 0; blk < MFMAS / 16; ++blk) {
            if constexpr (ISS == ISS_SPREAD) { piece(blk); }
            if constexpr (ISS == ISS_STAGGER) {
                // wave w issues its piece behind MFMA 2 (w & 7) of the block: eight waves, eight different points of the block
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    if (m == 2 * (wave & 7)) piece(blk);
                    acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % 12], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) acc[m % 12] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % 12], 0, 0, 0);
            }
            if constexpr (ISS == ISS_SPREAD || ISS == ISS_STAGGER) { if constexpr (PATH == PATH_REG) commit(blk); }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // consume one float of the landed stage so that nothing is dead
        a += smem[bufi * STAGE_F + ((tid * 7) & (STAGE_F - 1))] * 1e-30f;
    }
    if (tid == 0) out[blockIdx.x] = (float)(clock64() - t0) / (float)stages;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[blockIdx.x + 4096] = s;
}

template <int SRC, int PATH, int ISS, int MFMAS>
static void run(const char* name, const float* src, float* out) {
    const int blocks = 256, stages = 400;
    auto k = probe<SRC, PATH, ISS, MFMAS>;
    const size_t lds = 3 * 48 * 1024;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, src, out, stages);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipError_t err = hipGetLastError();
    std::vector<float> h(blocks);
    hipMemcpy(h.data(), out, blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = h[blocks / 2];
    const double bytes = PATH == PATH_NONE ? 0.0 : 48.0 * 1024;
    printf("%-64s %8.0f cycles/stage (min %7.0f max %7.0f)  %6.2f B/clk/CU  MFMA floor %5d  wall %7.3f ms  %s\n", name, cyc, h[0], h[blocks - 1],
           bytes / cyc, MFMAS * 2 * 32, ms, err == hipSuccess ? "" : hipGetErrorString(err));
    fflush(stdout);
}

int main() {
    const size_t n = (size_t)256 * 401 * 48 * 256;              // floats: 256 workgroups x 401 stages x 48 KiB = 4.7 GB
    float* src; float* out;
    if (hipMalloc(&src, n * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(src, 0, n * 4);
    hipMalloc(&out, 8192 * 4);
    for (int round = 0; round < 2; ++round) {
        printf("--- round %d\n", round);
        run<SRC_L2, PATH_NONE, ISS_BURST, 96>("no copies, 96 MFMAs per wave", src, out);
        run<SRC_L2, PATH_BUF, ISS_BURST, 0>("L2  buffer..lds   burst, no MFMAs", src, out);
        run<SRC_L2, PATH_GLDS, ISS_BURST, 0>("L2  global_load_lds burst, no MFMAs", src, out);
        run<SRC_L2, PATH_REG, ISS_BURST, 0>("L2  registers + ds_write burst, no MFMAs", src, out);
        run<SRC_HBM, PATH_BUF, ISS_BURST, 0>("HBM buffer..lds   burst, no MFMAs", src, out);
        run<SRC_HBM, PATH_REG, ISS_BURST, 0>("HBM registers + ds_write burst, no MFMAs", src, out);
        run<SRC_L2, PATH_BUF, ISS_BURST, 96>("L2  buffer..lds   burst, then 96 MFMAs", src, out);
        run<SRC_L2, PATH_BUF, ISS_SPREAD, 96>("L2  buffer..lds   one piece per 16 MFMAs, same point in all waves", src, out);
        run<SRC_L2, PATH_BUF, ISS_STAGGER, 96>("L2  buffer..lds   one piece per 16 MFMAs, staggered by wave", src, out);
        run<SRC_L2, PATH_GLDS, ISS_STAGGER, 96>("L2  global_load_lds one piece per 16 MFMAs, staggered by wave", src, out);
        run<SRC_L2, PATH_BUF, ISS_UPPER4, 96>("L2  buffer..lds   upper four waves burst 12 each, then 96 MFMAs", src, out);
        run<SRC_L2, PATH_REG, ISS_BURST, 96>("L2  registers + ds_write burst, then 96 MFMAs", src, out);
        run<SRC_L2, PATH_REG, ISS_STAGGER, 96>("L2  registers + ds_write one piece per 16 MFMAs, staggered", src, out);
        run<SRC_HBM, PATH_BUF, ISS_BURST, 96>("HBM buffer..lds   burst, then 96 MFMAs", src, out);
        run<SRC_HBM, PATH_BUF, ISS_SPREAD, 96>("HBM buffer..lds   one piece per 16 MFMAs, same point in all waves", src, out);
        run<SRC_HBM, PATH_BUF, ISS_STAGGER, 96>("HBM buffer..lds   one piece per 16 MFMAs, staggered by wave", src, out);
        run<SRC_HBM, PATH_REG, ISS_STAGGER, 96>("HBM registers + ds_write one piece per 16 MFMAs, staggered", src, out);
    }
    return 0;
}
