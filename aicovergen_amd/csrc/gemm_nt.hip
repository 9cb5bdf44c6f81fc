// "NT" GEMM on fp32 MFMA: C[r][o] = sum_i A[r][i] * W[o][i]  (both operands contiguous along the contraction).
// This is torch.nn.Linear applied to the LAST axis of a (B, C, T, F) map -- the time-distributed fully
// connected block ("TDF") of the MDX-Net separator, whose graph the reference runs through onnxruntime
// (reference src/mdx.py:74-77,193; architecture: kuielab TFC-TDF U-Net).
// Epilogue: v = acc + bias[o]; v = v * row_scale[ch] + row_shift[ch] (eval BatchNorm2d over the channel the row
// belongs to, ch = (r / rows_per_ch) % n_ch); v = act(v); v += res[r][o].
// Both tiles are staged through LDS with an odd row stride, so the strided fragment reads
// (lane -> row, fixed k) are bank-conflict free.
#include "common.h"

namespace aicg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* a;
    const float* w;
    const float* bias;
    const float* row_scale;
    const float* row_shift;
    const float* res;
    float* c;
    long R;
    int K, O;
    long lda, ldw, ldc, ldr;
    int rows_per_ch, n_ch, act;
};

static constexpr int GK = 32;        // K per stage
static constexpr int GLD = GK + 1;   // LDS row stride

// 128 x 128 tile, 4 waves as 2 x 2, each wave 64 x 64 (2 x 2 MFMA tiles)
__global__ void __launch_bounds__(256) gemm_nt_kernel(GemmArgs p) {
    __shared__ float As[128 * GLD];
    __shared__ float Ws[128 * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const long r0 = (long)blockIdx.x * 128;
    const int o0 = blockIdx.y * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = 0; k0 < p.K; k0 += GK) {
        __syncthreads();
        // stage 128 rows x 32 k of A and of W: consecutive lanes read consecutive k (coalesced 128-byte rows)
        for (int idx = tid; idx < 128 * GK; idx += 256) {
            const int row = idx >> 5, kk = idx & 31;
            const int k = k0 + kk;
            const long r = r0 + row;
            As[row * GLD + kk] = (r < p.R && k < p.K) ? p.a[r * p.lda + k] : 0.f;
            const int o = o0 + row;
            Ws[row * GLD + kk] = (o < p.O && k < p.K) ? p.w[(long)o * p.ldw + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < GK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(wm * 64 + i * 32 + l31) * GLD + kk + half];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Ws[(wn * 64 + j * 32 + l31) * GLD + kk + half];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // D layout: col (n = o) = lane & 31, row (m = r) = (reg & 3) + 8 * (reg >> 2) + 4 * half
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int o = o0 + wn * 64 + j * 32 + l31;
        if (o >= p.O) continue;
        const float bo = p.bias ? p.bias[o] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const long r = r0 + wm * 64 + i * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
                if (r >= p.R) continue;
                float v = acc[i][j][rg] + bo;
                if (p.row_scale) {
                    const int ch = (int)((r / p.rows_per_ch) % p.n_ch);
                    v = v * p.row_scale[ch] + p.row_shift[ch];
                }
                v = apply_act(v, p.act, 0.f);
                if (p.res) v += p.res[r * p.ldr + o];
                p.c[r * p.ldc + o] = v;
            }
    }
}

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_gemm_nt(const float* a, const float* w, const float* bias, const float* row_scale,
                            const float* row_shift, const float* res, float* c, int64_t R, int K, int O, int64_t lda,
                            int64_t ldw, int64_t ldc, int64_t ldr, int rows_per_ch, int n_ch, int act, void* stream) {
    if (!a || !w || !c) return fail(AICG_E_ARG, "aicg_gemm_nt: null pointer");
    if (R < 0 || K < 1 || O < 1) return fail(AICG_E_SHAPE, "aicg_gemm_nt: bad shape");
    if ((row_scale != nullptr) != (row_shift != nullptr) || (row_scale && (rows_per_ch < 1 || n_ch < 1)))
        return fail(AICG_E_ARG, "aicg_gemm_nt: row affine needs scale, shift, rows_per_ch and n_ch");
    if (R == 0) return AICG_OK;
    GemmArgs p{a, w, bias, row_scale, row_shift, res, c, (long)R, K, O, (long)lda, (long)ldw, (long)ldc, (long)ldr,
               rows_per_ch, n_ch, act};
    dim3 grid((unsigned)ldiv_up(R, 128), (unsigned)idiv_up(O, 128));
    hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("gemm_nt_kernel");
}
