// Micro-probe (GPU box only, not part of the library): the consumer k-step loop of conv_ws_kernel in isolation -- TM A fragments
// + TN B fragments read from LDS per k-step, TM*TN v_mfma_f32_32x32x2_f32 -- without producers, barriers or global traffic.
// Reports shader cycles per k-step per wave for several software-pipelining variants, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TM, int TN, int VARIANT>
__global__ void __launch_bounds__(256, 2) probe(float* out, unsigned long long* cyc, int ksteps, int reps, int bm, int chs) {
    extern __shared__ float smem[];
    float* ws = smem;                 // [k][bm]
    float* xs = smem + 68 * bm;       // [k][chs]
    for (int i = threadIdx.x; i < 68 * bm + 34 * chs; i += 256) smem[i] = (float)((i * 37) & 255) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* wt = ws + l31 + half * bm;
    const float* xt = xs + wave * 32 + l31 + half * chs;
    float a0[TM], b0[TN], a1[TM], b1[TN], a2[TM], b2[TN], a3[TM], b3[TN];
    auto fetch = [&](float (&a)[TM], float (&b)[TN], int s) {
        const int ss = s & 31;   // wrap inside the 64-row stage
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = wt[ss * 2 * bm + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = xt[(ss & 15) * 2 * chs + j * 32];
    };
    auto mma = [&](float (&a)[TM], float (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        if (VARIANT == 0) {            // distance-1 prefetch, two register sets (the shipped loop)
            fetch(a0, b0, 0);
            for (int s = 0; s + 2 <= ksteps; s += 2) {
                fetch(a1, b1, s + 1); mma(a0, b0);
                fetch(a0, b0, s + 2); mma(a1, b1);
            }
        } else if (VARIANT == 1) {     // distance-2 prefetch, four register sets
            fetch(a0, b0, 0); fetch(a1, b1, 1);
            for (int s = 0; s + 4 <= ksteps; s += 4) {
                fetch(a2, b2, s + 2); mma(a0, b0);
                fetch(a3, b3, s + 3); mma(a1, b1);
                fetch(a0, b0, s + 4); mma(a2, b2);
                fetch(a1, b1, s + 5); mma(a3, b3);
            }
        } else if (VARIANT == 2) {     // no LDS reads at all
            fetch(a0, b0, 0);
            for (int s = 0; s + 2 <= ksteps; s += 2) { mma(a0, b0); mma(a0, b0); }
        } else if (VARIANT == 4) {     // bursts: the 4 x (TM + TN) reads of four k-steps issued together, then 4 x TM x TN MFMAs back to back
            float A0[4][TM], B0[4][TN], A1[4][TM], B1[4][TN];
            auto fetch4 = [&](float (&A)[4][TM], float (&B)[4][TN], int s) {
#pragma unroll
                for (int u = 0; u < 4; ++u) fetch(A[u], B[u], s + u);
            };
            auto mma4 = [&](float (&A)[4][TM], float (&B)[4][TN]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) mma(A[u], B[u]);
            };
            fetch4(A0, B0, 0);
            for (int s = 0; s + 8 <= ksteps; s += 8) {
                fetch4(A1, B1, s + 4); mma4(A0, B0);
                fetch4(A0, B0, s + 8); mma4(A1, B1);
            }
        } else if (VARIANT == 5) {     // reads issued as in the shipped loop but the MFMAs use constant operands
            fetch(a0, b0, 0); fetch(a2, b2, 0);
            for (int s = 0; s + 2 <= ksteps; s += 2) {
                fetch(a1, b1, s + 1); mma(a2, b2);
                fetch(a0, b0, s + 2); mma(a2, b2);
            }
            acc[0][0][0] += a0[0] + b0[0] + a1[0] + b1[0];
        } else if (VARIANT == 6) {     // 16-byte fragments: [k/4][m][4] layouts, one ds_read_b128 per fragment per four k-steps
            float4 A0[TM], B0[TN], A1[TM], B1[TN];
            const float4* wt4 = reinterpret_cast<const float4*>(ws) + l31;            // [kgroup][bm] float4; lane half picks .xy / .zw
            const float4* xt4 = reinterpret_cast<const float4*>(xs) + wave * 32 + l31;
            auto fetchv = [&](float4 (&A)[TM], float4 (&B)[TN], int s) {
                const int g4 = (s >> 2) & 7;
#pragma unroll
                for (int i = 0; i < TM; ++i) A[i] = wt4[g4 * bm + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) B[j] = xt4[g4 * 160 + j * 32];
            };
            auto mmav = [&](float4 (&A)[TM], float4 (&B)[TN]) {
                float a[TM], b[TN];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = u == 0 ? A[i].x : u == 1 ? A[i].y : u == 2 ? A[i].z : A[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = u == 0 ? B[j].x : u == 1 ? B[j].y : u == 2 ? B[j].z : B[j].w;
                    mma(a, b);
                }
            };
            fetchv(A0, B0, 0);
            for (int s = 0; s + 8 <= ksteps; s += 8) {
                fetchv(A1, B1, s + 4); mmav(A0, B0);
                fetchv(A0, B0, s + 8); mmav(A1, B1);
            }
        } else if (VARIANT == 3) {     // distance-1, all reads of a step issued AFTER that step's first MFMA (reads under the MFMAs)
            fetch(a0, b0, 0);
            for (int s = 0; s + 2 <= ksteps; s += 2) {
                __builtin_amdgcn_sched_barrier(0);
                fetch(a1, b1, s + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                fetch(a0, b0, s + 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
            }
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
    const int bm = TM * 32, chs = 4 * 32 * TN + 1, ksteps = 432, reps = 20;
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : 70 * 1024;   // forces the residency
    hipFuncSetAttribute((const void*)probe<TM, TN, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * wgs_per_cu;
    for (int it = 0; it < 2; ++it) probe<TM, TN, V><<<grid, 256, lds>>>(out, cyc, ksteps, reps, bm, chs);
    hipDeviceSynchronize();
    static unsigned long long h[2048 * 4];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid * 4; ++i) s += (double)h[i];
    const double per = s / (grid * 4) / (double)(ksteps * reps);
    printf("%-34s %d WG/CU: %7.1f cycles per k-step per wave (MFMA floor %d; pipe busy %.0f %%)\n", name, wgs_per_cu, per, TM * TN * 64,
           100.0 * wgs_per_cu * TM * TN * 64 / per);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 4 * 8);
    for (int w = 1; w <= 2; ++w) {
        run<3, 1, 2>("TM3 TN1 no LDS reads", w, out, cyc);
        run<3, 1, 0>("TM3 TN1 distance-1 (shipped)", w, out, cyc);
        run<3, 1, 1>("TM3 TN1 distance-2", w, out, cyc);
        run<3, 1, 3>("TM3 TN1 distance-1 sched-fenced", w, out, cyc);
        run<3, 1, 4>("TM3 TN1 bursts of 4 k-steps", w, out, cyc);
        run<3, 1, 5>("TM3 TN1 reads unused by the MFMAs", w, out, cyc);
        run<3, 1, 6>("TM3 TN1 ds_read_b128 fragments", w, out, cyc);
        run<2, 1, 4>("TM2 TN1 bursts of 4 k-steps", w, out, cyc);
        run<2, 1, 6>("TM2 TN1 ds_read_b128 fragments", w, out, cyc);
        run<1, 2, 6>("TM1 TN2 ds_read_b128 fragments", w, out, cyc);
        run<2, 1, 0>("TM2 TN1 distance-1", w, out, cyc);
        run<2, 1, 1>("TM2 TN1 distance-2", w, out, cyc);
        run<1, 2, 0>("TM1 TN2 distance-1", w, out, cyc);
        run<2, 2, 0>("TM2 TN2 distance-1", w, out, cyc);
        run<4, 1, 0>("TM4 TN1 distance-1", w, out, cyc);
        run<5, 1, 0>("TM5 TN1 distance-1", w, out, cyc);
    }
    return 0;
}
