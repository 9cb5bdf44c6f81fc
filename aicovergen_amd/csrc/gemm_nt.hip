// "NT" GEMM on fp32 MFMA: C[r][o] = sum_i A[r][i] * W[o][i]  (both operands contiguous along the contraction).
// This is torch.nn.Linear applied to the LAST axis of a (B, C, T, F) map -- the time-distributed fully
// connected block ("TDF") of the MDX-Net separator, whose graph the reference runs through onnxruntime
// (reference src/mdx.py:74-77,193; architecture: kuielab TFC-TDF U-Net).
// Epilogue: v = acc + bias[o]; v = v * row_scale[ch] + row_shift[ch] (eval BatchNorm2d over the channel the row
// belongs to, ch = (r / rows_per_ch) % n_ch); v = act(v); v += res[r][o].
// Both tiles are staged through LDS with an odd row stride, so the strided fragment reads
// (lane -> row, fixed k) are bank-conflict free.
#include <type_traits>

#include "common.h"

#include <cstdint>
#include <cstdlib>

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
    int order;
    int wide;   // c, res (and bias) are 16-byte aligned with row strides that are multiples of 4: interior tiles may use float4
};

static constexpr int GK = 32;        // K per stage
static constexpr int GLD = GK + 1;   // LDS row stride

// Epilogue shared by the fp32 and the split-precision kernels.  `scratch` = this wave's 32 x 36 floats of (free) LDS; the caller
// has synchronised the workgroup after its last K stage.
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[2][2], long r0, int o0, int wm, int wn, int lane,
                                              float* scratch) {
    const int half = lane >> 5, l31 = lane & 31;
    // D layout: col (n = o) = lane & 31, row (m = r) = (reg & 3) + 8 * (reg >> 2) + 4 * half.
    // The per-channel affine (eval BatchNorm2d) is constant over a 32-row MFMA tile when rows_per_ch is a multiple of 32 (the MDX
    // maps have rows_per_ch = dim_t = 256): one division per tile instead of one per element.
    const bool ch_per_tile = p.row_scale && (p.rows_per_ch & 31) == 0;
    auto epilogue = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = o0 + wn * 64 + j * 32 + l31;
            if (o >= p.O) continue;
            const float bo = p.bias ? p.bias[o] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long rt = r0 + wm * 64 + i * 32;
                float sc = 1.f, sh = 0.f;
                if (ch_per_tile) {
                    const int ch = (int)((rt / p.rows_per_ch) % p.n_ch);
                    sc = p.row_scale[ch]; sh = p.row_shift[ch];
                }
                float rv[16];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const long r = rt + (rg & 3) + 8 * (rg >> 2) + 4 * half;
                    rv[rg] = (p.res && r < p.R) ? p.res[r * p.ldr + o] : 0.f;
                }
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const long r = rt + (rg & 3) + 8 * (rg >> 2) + 4 * half;
                    if (r >= p.R) continue;
                    float v = acc[i][j][rg] + bo;
                    if (p.row_scale && !ch_per_tile) {
                        const int ch = (int)((r / p.rows_per_ch) % p.n_ch);
                        sc = p.row_scale[ch]; sh = p.row_shift[ch];
                    }
                    v = v * sc + sh;
                    v = ACT == 0 ? v : ACT == 1 ? (v > 0.f ? v : 0.f) : apply_act(v, p.act, 0.f);
                    p.c[r * p.ldc + o] = v + rv[rg];
                }
            }
        }
    };
    // Interior tiles with 16-byte-aligned rows: float4 epilogue.  A lane owns one column o and 16 rows per 32 x 32 tile -- 16 dword
    // stores (+ 16 residual loads) per tile, and the tail of a tile is bound by the NUMBER of memory instructions (the f/8 -> f
    // expansion has only 12 K stages per tile against 64 + 64 of them).  Each tile takes a detour through a per-wave LDS scratch
    // that turns the layout into 4 consecutive columns per lane: 4 float4 stores (8 rows x 128 B each) and 4 float4 residual loads.
    const bool wide = p.wide && r0 + 128 <= p.R && o0 + 128 <= p.O && ch_per_tile == (p.row_scale != nullptr);
    if (wide) {
        constexpr int SR = 36;
        float* wr = scratch + (4 * half) * SR + l31;
        const int rrow = lane >> 3, rcol = (lane & 7) * 4;
        const float4* rd = reinterpret_cast<const float4*>(scratch + rrow * SR + rcol);
        auto wide_body = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = o0 + wn * 64 + j * 32 + rcol;
                float4 bo = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) bo = *reinterpret_cast<const float4*>(p.bias + o);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const long rt = r0 + wm * 64 + i * 32;
                    float sc = 1.f, sh = 0.f;
                    if (p.row_scale) {
                        const int ch = (int)((rt / p.rows_per_ch) % p.n_ch);
                        sc = p.row_scale[ch]; sh = p.row_shift[ch];
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) wr[((rg & 3) + 8 * (rg >> 2)) * SR] = acc[i][j][rg];
                    __builtin_amdgcn_wave_barrier();
                    float4 rv[4], v[4];
                    if (p.res) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const float4*>(p.res + (rt + rrow + 8 * q) * p.ldr + o);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = rd[q * 8 * (SR / 4)];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float e[4] = {v[q].x + bo.x, v[q].y + bo.y, v[q].z + bo.z, v[q].w + bo.w};
                        const float rr[4] = {p.res ? rv[q].x : 0.f, p.res ? rv[q].y : 0.f, p.res ? rv[q].z : 0.f, p.res ? rv[q].w : 0.f};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            float x = e[t] * sc + sh;
                            x = ACT == 0 ? x : ACT == 1 ? (x > 0.f ? x : 0.f) : apply_act(x, p.act, 0.f);
                            e[t] = x + rr[t];
                        }
                        *reinterpret_cast<float4*>(p.c + (rt + rrow + 8 * q) * p.ldc + o) = make_float4(e[0], e[1], e[2], e[3]);
                    }
                }
            }
        };
        if (p.act == AICG_ACT_NONE) wide_body(std::integral_constant<int, 0>{});
        else if (p.act == AICG_ACT_RELU) wide_body(std::integral_constant<int, 1>{});
        else wide_body(std::integral_constant<int, 2>{});
        return;
    }
    if (p.act == AICG_ACT_NONE) epilogue(std::integral_constant<int, 0>{});
    else if (p.act == AICG_ACT_RELU) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 2>{});
}

// 128 x 128 tile, 4 waves as 2 x 2, each wave 64 x 64 (2 x 2 MFMA tiles)
__global__ void __launch_bounds__(256) gemm_nt_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float As[128 * GLD + 4];
    __shared__ __attribute__((aligned(16))) float Ws[128 * GLD + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    // tile order (order == 1): column tiles fastest inside groups of 8 consecutive ids, groups dealt to the XCDs -- the workgroups
    // that share a row tile of A run back to back behind one L2 instead of re-streaming A once per column tile
    long r0;
    int o0;
    if (p.order == 0) {
        r0 = (long)blockIdx.x * 128;
        o0 = blockIdx.y * 128;
    } else {
        const unsigned nx = gridDim.x, ny = gridDim.y;
        const unsigned flat = blockIdx.y * nx + blockIdx.x;
        const unsigned tile = xcd_remap(flat, nx * ny);
        r0 = (long)(tile / ny) * 128;
        o0 = (int)(tile % ny) * 128;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Software pipeline as in conv.hip: the float4 global loads of K-slab s+1 are issued right after the barrier that
    // publishes slab s and land while its MFMAs run; fragments alternate between two register sets.
    float4 av[4], wv[4];
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;          // 128 rows x 8 float4
            const int row = idx >> 3, k = k0 + (idx & 7) * 4;
            const long r = r0 + row;
            const int o = o0 + row;
            av[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            wv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < p.R && k < p.K) av[e] = *reinterpret_cast<const float4*>(p.a + r * p.lda + k);
            if (o < p.O && k < p.K) wv[e] = *reinterpret_cast<const float4*>(p.w + (long)o * p.ldw + k);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int row = idx >> 3, kk = (idx & 7) * 4;
            float* pa = As + row * GLD + kk;
            float* pw = Ws + row * GLD + kk;
            pa[0] = av[e].x; pa[1] = av[e].y; pa[2] = av[e].z; pa[3] = av[e].w;
            pw[0] = wv[e].x; pw[1] = wv[e].y; pw[2] = wv[e].z; pw[3] = wv[e].w;
        }
    };
    const float* ap = As + (wm * 64 + l31) * GLD + half;
    const float* wp = Ws + (wn * 64 + l31) * GLD + half;
    prefetch(0);
    for (int k0 = 0; k0 < p.K; k0 += GK) {
        __syncthreads();
        commit();
        __syncthreads();
        if (k0 + GK < p.K) prefetch(k0 + GK);
        float a0[2], b0[2], a1[2], b1[2];
        auto fetch = [&](float (&a)[2], float (&b)[2], int kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = ap[i * 32 * GLD + kk];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = wp[j * 32 * GLD + kk];
        };
        auto mma = [&](float (&a)[2], float (&b)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        };
        fetch(a0, b0, 0);
#pragma unroll
        for (int kk = 0; kk < GK; kk += 4) {
            fetch(a1, b1, kk + 2);
            mma(a0, b0);
            fetch(a0, b0, kk + 4);  // kk + 4 == GK reads the (unused) pad column of the next row: in bounds
            mma(a1, b1);
        }
    }
    __syncthreads();   // every wave is done reading the last K stage: As / Ws are free (2 x 1152 scratch floats in each)
    gemm_epilogue(p, acc, r0, o0, wm, wn, lane, (wave < 2 ? As : Ws) + (wave & 1) * (32 * 36));
}

// ---- opt-in split precision (aicg_gemm_nt_split; see conv_ws3s.h for the arithmetic) ------------------------------------------
// Same tile and epilogue; operands are split into bf16 hi / lo while they are committed to LDS and contracted as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16.  K runs contiguously in both operands, so an item = 8 consecutive k of one
// row = two float4 loads -> one hi and one lo 16-byte word, stored [k-chunk][hi|lo][row] (129-word planes: the eight threads that
// share a row land in distinct bank groups); lane (l31, half) of k16 step s reads chunk 2 s + half of row l31: one ds_read_b128
// per fragment and part, no conflicts.
static constexpr int SK = 64;          // K per stage
static constexpr int SPL = 129;        // float4 words per (chunk, part) plane
static constexpr int SOP = (SK / 8) * 2 * SPL;   // float4 words per operand stage

__global__ void __launch_bounds__(256, 2) gemm_nt_split_kernel(GemmArgs p) {
    HIP_DYNAMIC_SHARED(float4, smem4)
    float4* const As = smem4;
    float4* const Ws = smem4 + SOP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    long r0;
    int o0;
    if (p.order == 0) {
        r0 = (long)blockIdx.x * 128;
        o0 = blockIdx.y * 128;
    } else {
        const unsigned nx = gridDim.x, ny = gridDim.y;
        const unsigned flat = blockIdx.y * nx + blockIdx.x;
        const unsigned tile = xcd_remap(flat, nx * ny);
        r0 = (long)(tile / ny) * 128;
        o0 = (int)(tile % ny) * 128;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 av[4][2], wv[4][2];   // 4 items (row, 8-k chunk) per operand and thread
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;          // 128 rows x 8 chunks
            const int row = idx >> 3, k = k0 + (idx & 7) * 8;
            const long r = r0 + row;
            const int o = o0 + row;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                av[e][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                wv[e][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < p.R && k + 4 * h < p.K) av[e][h] = *reinterpret_cast<const float4*>(p.a + r * p.lda + k + 4 * h);
                if (o < p.O && k + 4 * h < p.K) wv[e][h] = *reinterpret_cast<const float4*>(p.w + (long)o * p.ldw + k + 4 * h);
            }
        }
    };
    auto split8 = [](const float4 (&v)[2], float4& hi, float4& lo) {
        unsigned h[4], l[4];
        split_bf16_pair(v[0].x, v[0].y, h[0], l[0]);
        split_bf16_pair(v[0].z, v[0].w, h[1], l[1]);
        split_bf16_pair(v[1].x, v[1].y, h[2], l[2]);
        split_bf16_pair(v[1].z, v[1].w, h[3], l[3]);
        hi = make_float4(__builtin_bit_cast(float, h[0]), __builtin_bit_cast(float, h[1]), __builtin_bit_cast(float, h[2]),
                         __builtin_bit_cast(float, h[3]));
        lo = make_float4(__builtin_bit_cast(float, l[0]), __builtin_bit_cast(float, l[1]), __builtin_bit_cast(float, l[2]),
                         __builtin_bit_cast(float, l[3]));
    };
    auto commit = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int row = idx >> 3, chunk = idx & 7;
            float4 hi, lo;
            split8(av[e], hi, lo);
            As[(chunk * 2) * SPL + row] = hi;
            As[(chunk * 2 + 1) * SPL + row] = lo;
            split8(wv[e], hi, lo);
            Ws[(chunk * 2) * SPL + row] = hi;
            Ws[(chunk * 2 + 1) * SPL + row] = lo;
        }
    };
    const float4* ap = As + (2 * half) * SPL + wm * 64 + l31;
    const float4* wp = Ws + (2 * half) * SPL + wn * 64 + l31;
    prefetch(0);
    for (int k0 = 0; k0 < p.K; k0 += SK) {
        __syncthreads();
        commit();
        __syncthreads();
        if (k0 + SK < p.K) prefetch(k0 + SK);
        struct Frag { float4 ah[2], al[2], bh[2], bl[2]; };
        auto fetch = [&](Frag& f, int s) {   // k16 step s: chunks 2 s, 2 s + 1
#pragma unroll
            for (int i = 0; i < 2; ++i) { f.ah[i] = ap[s * 4 * SPL + i * 32]; f.al[i] = ap[s * 4 * SPL + SPL + i * 32]; }
#pragma unroll
            for (int j = 0; j < 2; ++j) { f.bh[j] = wp[s * 4 * SPL + j * 32]; f.bl[j] = wp[s * 4 * SPL + SPL + j * 32]; }
        };
        auto mma = [&](Frag& f) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = mfma_bf16_32x32x16(f.al[i], f.bh[j], acc[i][j]);
                    acc[i][j] = mfma_bf16_32x32x16(f.ah[i], f.bl[j], acc[i][j]);
                    acc[i][j] = mfma_bf16_32x32x16(f.ah[i], f.bh[j], acc[i][j]);
                }
        };
        Frag f0, f1;
        fetch(f0, 0);
        fetch(f1, 1);
        mma(f0);
        fetch(f0, 2);
        mma(f1);
        fetch(f1, 3);
        mma(f0);
        mma(f1);
    }
    __syncthreads();
    gemm_epilogue(p, acc, r0, o0, wm, wn, lane, reinterpret_cast<float*>(smem4) + wave * (32 * 36));
}

}  // namespace aicg

using namespace aicg;

static int gemm_nt_launch(bool split, const float* a, const float* w, const float* bias, const float* row_scale,
                          const float* row_shift, const float* res, float* c, int64_t R, int K, int O, int64_t lda,
                          int64_t ldw, int64_t ldc, int64_t ldr, int rows_per_ch, int n_ch, int act, void* stream) {
    if (!a || !w || !c) return fail(AICG_E_ARG, "aicg_gemm_nt: null pointer");
    if (R < 0 || K < 1 || O < 1) return fail(AICG_E_SHAPE, "aicg_gemm_nt: bad shape");
    if ((row_scale != nullptr) != (row_shift != nullptr) || (row_scale && (rows_per_ch < 1 || n_ch < 1)))
        return fail(AICG_E_ARG, "aicg_gemm_nt: row affine needs scale, shift, rows_per_ch and n_ch");
    if ((K & 3) || (lda & 3) || (ldw & 3))
        return fail(AICG_E_SHAPE, "aicg_gemm_nt: K, lda and ldw must be multiples of 4 (float4 loads)");
    if (R == 0) return AICG_OK;
    GemmArgs p{a, w, bias, row_scale, row_shift, res, c, (long)R, K, O, (long)lda, (long)ldw, (long)ldc, (long)ldr,
               rows_per_ch, n_ch, act, 0, 0};
    AICG_SWITCH(order, "AICG_GEMM_ORDER", 1);
    AICG_SWITCH(wide, "AICG_GEMM_WIDE", 1);
    p.order = order;
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    p.wide = wide && al(c) && (ldc & 3) == 0 && (!res || (al(res) && (ldr & 3) == 0)) && (!bias || al(bias)) && (O & 3) == 0;
    dim3 grid((unsigned)ldiv_up(R, 128), (unsigned)idiv_up(O, 128));
    if (split) {
        const size_t lds = (size_t)2 * SOP * sizeof(float4);
        allow_dynamic_lds((const void*)gemm_nt_split_kernel, lds);
        hipLaunchKernelGGL(gemm_nt_split_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
        return check_launch("gemm_nt_split_kernel");
    }
    hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("gemm_nt_kernel");
}

extern "C" int aicg_gemm_nt(const float* a, const float* w, const float* bias, const float* row_scale,
                            const float* row_shift, const float* res, float* c, int64_t R, int K, int O, int64_t lda,
                            int64_t ldw, int64_t ldc, int64_t ldr, int rows_per_ch, int n_ch, int act, void* stream) {
    return gemm_nt_launch(false, a, w, bias, row_scale, row_shift, res, c, R, K, O, lda, ldw, ldc, ldr, rows_per_ch, n_ch, act, stream);
}

extern "C" int aicg_gemm_nt_split(const float* a, const float* w, const float* bias, const float* row_scale,
                                  const float* row_shift, const float* res, float* c, int64_t R, int K, int O, int64_t lda,
                                  int64_t ldw, int64_t ldc, int64_t ldr, int rows_per_ch, int n_ch, int act, void* stream) {
    return gemm_nt_launch(true, a, w, bias, row_scale, row_shift, res, c, R, K, O, lda, ldw, ldc, ldr, rows_per_ch, n_ch, act, stream);
}
