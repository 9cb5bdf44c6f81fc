// Micro-probe: consumer loop of a split-precision (bf16 hi/lo, 3 MFMAs per product) conv tile -- LDS b128 fragment reads + v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int TM, int TN, int VARIANT>
__global__ void __launch_bounds__(256, 2) probe(float* out, unsigned long long* cyc, int ksteps, int reps, int bm, int chs) {
    extern __shared__ float smem[];
    for (int i = threadIdx.x; i < 24 * 1024; i += 256) smem[i] = (float)((i * 37) & 255) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float4* wt = reinterpret_cast<const float4*>(smem) + l31 + half * bm;                // [g][hi/lo][half][bm] float4 units
    const float4* xt = reinterpret_cast<const float4*>(smem) + 3072 + wave * 32 + l31 + half * chs;
    float4 ah0[TM], al0[TM], bh0[TN], bl0[TN], ah1[TM], al1[TM], bh1[TN], bl1[TN];
    auto fetch = [&](float4 (&ah)[TM], float4 (&al)[TM], float4 (&bh)[TN], float4 (&bl)[TN], int s) {
        const int g = s & 3;
#pragma unroll
        for (int i = 0; i < TM; ++i) { ah[i] = wt[g * 4 * bm + i * 32]; al[i] = wt[g * 4 * bm + 2 * bm + i * 32]; }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bh[j] = xt[g * 4 * chs + j * 32]; bl[j] = xt[g * 4 * chs + 2 * chs + j * 32]; }
    };
    auto mma = [&](float4 (&ah)[TM], float4 (&al)[TM], float4 (&bh)[TN], float4 (&bl)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const bf16x8 AH = __builtin_bit_cast(bf16x8, ah[i]), AL = __builtin_bit_cast(bf16x8, al[i]);
                const bf16x8 BH = __builtin_bit_cast(bf16x8, bh[j]), BL = __builtin_bit_cast(bf16x8, bl[j]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, acc[i][j], 0, 0, 0);
            }
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        if (VARIANT == 0) {
            fetch(ah0, al0, bh0, bl0, 0);
            for (int s = 0; s + 2 <= ksteps; s += 2) {
                fetch(ah1, al1, bh1, bl1, s + 1); mma(ah0, al0, bh0, bl0);
                fetch(ah0, al0, bh0, bl0, s + 2); mma(ah1, al1, bh1, bl1);
            }
        } else {   // MFMAs only
            fetch(ah0, al0, bh0, bl0, 0);
            for (int s = 0; s + 2 <= ksteps; s += 2) { mma(ah0, al0, bh0, bl0); mma(ah0, al0, bh0, bl0); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int TM, int TN, int V>
static void run(const char* name, int wgs_per_cu, float* out, unsigned long long* cyc) {
    const int bm = TM * 32, chs = 4 * 32 * TN + 1, ksteps = 54, reps = 100;   // 54 k16 steps = K 864
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : 76 * 1024;
    hipFuncSetAttribute((const void*)probe<TM, TN, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * wgs_per_cu;
    for (int it = 0; it < 2; ++it) probe<TM, TN, V><<<grid, 256, lds>>>(out, cyc, ksteps, reps, bm, chs);
    hipDeviceSynchronize();
    static unsigned long long h[2048 * 4];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid * 4; ++i) s += (double)h[i];
    const double per = s / (grid * 4) / (double)(ksteps * reps);
    printf("%-30s %d WG/CU: %7.1f cycles per K16 step per wave (MFMA floor %d; fp32 path floor for the same K: %d)\n", name, wgs_per_cu, per,
           TM * TN * 3 * 32, TM * TN * 8 * 64);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 4 * 8);
    for (int w = 1; w <= 2; ++w) {
        run<3, 1, 1>("TM3 TN1 MFMA only", w, out, cyc);
        run<3, 1, 0>("TM3 TN1 with b128 fragments", w, out, cyc);
        run<2, 2, 0>("TM2 TN2 with b128 fragments", w, out, cyc);
        run<4, 1, 0>("TM4 TN1 with b128 fragments", w, out, cyc);
        run<1, 2, 0>("TM1 TN2 with b128 fragments", w, out, cyc);
    }
    return 0;
}
