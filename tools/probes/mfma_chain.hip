// How much does a v_mfma_f32_32x32x2_f32 / 16x16x4 that accumulates onto the result of the one right before it cost, against the same
// number of MFMAs spread over 2 / 4 / 12 accumulators?  (tdf_pair's phase 2 is one 96-long chain per stage, phase 1 chains of 4.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_chain tools/probes/mfma_chain.hip && tools/probes/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int RUN>   // NACC accumulators, RUN consecutive MFMAs on one before moving to the next
__global__ void __launch_bounds__(256) chain32(float* out, int iters, float seed) {
    f32x16 a[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) a[i][r] = seed + i;
    float x = seed * (float)(threadIdx.x & 7) + 0.5f, y = 0.25f + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 48 / (NACC * RUN); ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int r = 0; r < RUN; ++r) a[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += a[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int RUN>
__global__ void __launch_bounds__(256) chain16(float* out, int iters, float seed) {
    f32x4 a[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) a[i][r] = seed + i;
    float x = seed * (float)(threadIdx.x & 7) + 0.5f, y = 0.25f + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 48 / (NACC * RUN); ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int r = 0; r < RUN; ++r) a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += a[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K k, float* out, double flop_per_mfma) {
    const int blocks = 256, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-9f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * 48;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s  (one wave per SIMD, 256 workgroups)\n", name, ms, n * flop_per_mfma / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    const double f32 = 2.0 * 32 * 32 * 2, f16 = 2.0 * 16 * 16 * 4;
    for (int round = 0; round < 2; ++round) {
        run("32x32x2  1 acc (chain)", chain32<1, 48>, out, f32);
        run("32x32x2  2 acc alternating", chain32<2, 1>, out, f32);
        run("32x32x2  4 acc alternating", chain32<4, 1>, out, f32);
        run("32x32x2 12 acc, runs of 4", chain32<12, 4>, out, f32);
        run("32x32x2 12 acc alternating", chain32<12, 1>, out, f32);
        run("32x32x2  2 acc, runs of 4", chain32<2, 4>, out, f32);
        run("16x16x4  1 acc (chain)", chain16<1, 48>, out, f16);
        run("16x16x4  2 acc alternating", chain16<2, 1>, out, f16);
        run("16x16x4  4 acc alternating", chain16<4, 1>, out, f16);
        run("16x16x4 48 acc alternating", chain16<48, 1>, out, f16);
    }
    return 0;
}
