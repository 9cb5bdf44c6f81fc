// The TDF block of the MDX-Net separator as ONE kernel:
//     out = x + relu(bn2(relu(bn1(x W1^T + b1)) W2^T + b2))        x, out: (R, F) rows of a (B, C, T, F) map, W1: (H, F), W2: (F, H), H = F / bn
// (reference: the graph onnxruntime executes at src/mdx.py:74-77,193; architecture: kuielab TFC_TDF).
//
// Two launches of the NT GEMM (gemm_nt.hip) ran this pair at 103 TFLOP/s (r2): the F -> H contraction re-reads x once per 128-column
// tile of H (FETCH x 3.05), the (R, H) intermediate makes an HBM round trip, and the H -> F expansion has only H / 32 K-stages per
// tile against a full tile epilogue.  Here a workgroup owns 128 rows:
//   phase 1  four waves, 32 rows each, accumulate ALL H columns of their rows (H / 32 accumulator tiles per wave: up to 192 registers --
//            one wave per SIMD, so the whole 512-entry register file is the wave's) over K-stages of 32 staged through LDS: W1 as
//            [k / 8][k & 1][h][4] quads (its packed image in HBM has that order: float4 copies), x rows as [k / 8][k & 1][row][4]
//            (two float4 loads -> two ds_write_b128 per 8 k of a row).  One ds_read_b128 per fragment per four MFMA k-steps,
//            H / 32 + 1 fragments per 4 H / 32 MFMAs.  A stage is H / 8 MFMAs of 64 cycles per wave (12 288 cycles at H = 384): the
//            same waves issue the next stage's 16 global loads at its start and commit them behind the barrier that ends it -- no
//            producer waves, their registers would halve the budget of the accumulators.
//   hand-over bias, BatchNorm (per row channel), ReLU in registers -- the intermediate NEVER leaves the register file:
//   phase 2  out^T tile (32 f x 32 rows) += W2 tile (from LDS) x intermediate^T, with the phase-1 ACCUMULATOR REGISTERS as the B operand:
//            register 4 q + e of tile i holds h = 32 i + 8 q + 4 half + e for this lane's row, so MFMA step (i, q, e) contracts the pair
//            (h, h + 4) and lane half `half` of the A operand supplies W2[f][h + 4 half] -- four consecutive h: one ds_read_b128 of a
//            row-major W2 row per four steps.  One LDS stage = 32 output columns; its epilogue (bias, BatchNorm, ReLU,
//            + x, float4 stores: a lane owns 4 consecutive f per register quad) runs behind H / 8 MFMAs.
// x is read twice (phase 1 operand, phase 2 residual), out written once, nothing else touches HBM but the L2-resident weights.
#include "common.h"

#include <cstdint>

namespace aicg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TdfArgs {
    const float* x;
    const float* w1p;   // packed: [F / 8][2][H][4], element e of quad (g, par, h) = W1[h][8 g + 2 e + par]
    const float* b1;
    const float* s1;    // eval BatchNorm2d over the channel a row belongs to: ch = (row / rows_per_ch) % n_ch
    const float* t1;
    const float* w2p;   // packed: [F / 32] slabs of tdf_w2_slab_floats(H): 32 rows of H + 4 floats (W2 rows, zero padded)
    const float* b2;
    const float* s2;
    const float* t2;
    float* out;
    long R;
    int F, H, rows_per_ch, n_ch;
};

static constexpr int TR = 128;   // rows per workgroup
static constexpr int TK = 32;    // K per phase-1 stage

// floats of one packed W2 slab: 32 rows of H + 4 (row padding against LDS bank conflicts), rounded up to whole 1 KiB DMA pieces
__host__ __device__ constexpr int tdf_w2_slab_floats(int H) { return (32 * (H + 4) + 255) / 256 * 256; }

// LDS floats per stage buffer
__host__ __device__ constexpr int tdf_stage_floats(int H) {
    const int p1 = TK * H + TK * TR;          // W1 slab (8 planes x H quads) + x slab (8 planes x 128 quads)
    const int p2 = tdf_w2_slab_floats(H);     // W2 slab: 32 rows of H + 4
    return p1 > p2 ? p1 : p2;
}

// Staging (all 256 threads).  Stage st < n1: K-slab st of phase 1 (W1 quads + x quads); stage n1 + fb: W2 rows of output block fb.
// Buffer st & 1.  The weight slabs are plain contiguous copies -- both images are laid out in HBM exactly as LDS wants them (W2 with
// its row padding materialised) -- and go HBM -> LDS by DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, lane-linear
// destination, no staging registers: the accumulators need them); only the x slab, whose 8 consecutive k per row are dealt to
// two parity planes, passes through registers (four float4).
__device__ __forceinline__ void lds_dma16(const float* g, float* lds_wave_base, int lane) {
#ifdef AICG_EMULATED
    *reinterpret_cast<float4*>(lds_wave_base + 4 * lane) = *reinterpret_cast<const float4*>(g);
#else
    (void)lane;
    // Issued through inline asm ON PURPOSE.  With __builtin_amdgcn_global_load_lds the compiler knows the instruction writes LDS, cannot
    // prove that the destination (the OTHER stage buffer) does not alias the fragment reads that follow, and puts an
    // `s_waitcnt vmcnt(0)` in front of the first ds_read: every stage then exposes a full HBM round trip before its MFMAs (measured r3:
    // 10.2 ms for the level-0 block, of which 3.9 ms remained with every MFMA removed).  The asm form is invisible to that pass; the
    // hand-written dma_wait() at the end of the stage is what orders it.  M0 carries the wave-uniform LDS base and is restored.
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds)
                 : "memory");
#endif
}
__device__ __forceinline__ void sched_fence() {
#ifndef AICG_EMULATED
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// every DMA this wave issued has landed (the LDS-only barrier behind it publishes the stage)
__device__ __forceinline__ void dma_wait() {
#ifndef AICG_EMULATED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// n4 float4 (a multiple of 64) from src to dst, spread over the workgroup's four waves
__device__ __forceinline__ void tdf_dma_slab(const float* src, float* dst, int n4, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int c = wave * 64; c < n4; c += 256) lds_dma16(src + 4 * (c + lane), dst + 4 * c, lane);
}

// piece k of this wave's share of a slab (its pieces are c = wave * 64 + k * 256 < n4): ONE 1 KiB DMA instruction.  The stages issue
// their next slab a piece at a time between blocks of four MFMAs: a burst of 12-13 pieces per wave at the top of a stage holds the
// wave -- the only one on its SIMD -- in the issue of the DMA instructions while the CU's LDS-DMA path takes them at ~16 B per clock
// (ablation, profiles/r05_tdf_pair_ablation.txt: 0.64 + 0.34 ms of the level-0 block's 7.59 were these bursts).
// Eight instructions per piece: the source as an SGPR base (slab + 4 KiB x k, scalar adds) + ONE per-thread byte offset that never
// changes (16 x thread id: c + lane = tid + 256 k), the LDS destination as scalar arithmetic on the stage buffer's LDS address (no
// generic-to-LDS pointer cast with its null check per piece), a range check only for the slab's ragged last piece (n4_all = float4
// count every one of this call's pieces k < n4_all / 256 is inside for all four waves).  The first form cost ~20 issue slots per piece.
__device__ __forceinline__ void tdf_dma_piece(const float* src, float* dst, unsigned dst_lds, int n4, int tid, unsigned voff16, int wave, int k) {
    const int c = wave * 64 + k * 256;
    if (256 * k + 192 >= n4 && c >= n4) return;            // (the first clause folds at compile time where n4 is a constant)
#ifdef AICG_EMULATED
    (void)dst_lds; (void)voff16;
    const int lane = tid & 63;
    *reinterpret_cast<float4*>(dst + 4 * c + 4 * lane) = *reinterpret_cast<const float4*>(src + 4 * (c + lane));
#else
    (void)dst;
    const char* sp = reinterpret_cast<const char*>(src) + 4096 * k;
    const unsigned lds = dst_lds + 1024u * (unsigned)wave + 4096u * (unsigned)k;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff16), "s"(sp), "s"(lds)
                 : "memory");
#endif
}

struct XStage { float4 a0, a1, b0, b1; };   // two (row, 8-k group) items of the x slab

__device__ __forceinline__ void tdf_load_x(const TdfArgs& p, int tid, long r0, int st, XStage& v) {
    // item = (8-k group, row); rows fastest: consecutive LDS quads.  Items tid and tid + 256: groups gg and gg + 2 of row tid & 127
    const int row = tid & (TR - 1), gg = tid >> 7;
    const long r = r0 + row;
    const bool ok = r < p.R;
    const float4* xr = reinterpret_cast<const float4*>(p.x + (ok ? r : 0) * p.F + (long)st * TK + 8 * gg);
    // raw values only: anything that CONSUMES them here (even the zero select of the rows past R) would be scheduled right behind the
    // loads and wait for them in front of the stage's MFMAs -- tdf_commit_x masks
    v.a0 = xr[0]; v.a1 = xr[1]; v.b0 = xr[4]; v.b1 = xr[5];   // + 16 floats: group gg + 2
}

__device__ __forceinline__ void tdf_commit_x(const TdfArgs& p, float* xbuf, int tid, long r0, XStage v) {
    const int row = tid & (TR - 1), gg = tid >> 7;
    if (r0 + row >= p.R) v.a0 = v.a1 = v.b0 = v.b1 = make_float4(0.f, 0.f, 0.f, 0.f);   // rows past the end read row 0: zero them
    float4* xdst = reinterpret_cast<float4*>(xbuf);
    // k = 8 g + 2 j + par: parity 0 takes elements 0, 2, 4, 6 of the 8 loaded values, parity 1 the odd ones
    xdst[(gg * 2) * TR + row] = make_float4(v.a0.x, v.a0.z, v.a1.x, v.a1.z);
    xdst[(gg * 2 + 1) * TR + row] = make_float4(v.a0.y, v.a0.w, v.a1.y, v.a1.w);
    xdst[((gg + 2) * 2) * TR + row] = make_float4(v.b0.x, v.b0.z, v.b1.x, v.b1.z);
    xdst[((gg + 2) * 2 + 1) * TR + row] = make_float4(v.b0.y, v.b0.w, v.b1.y, v.b1.w);
}

// ABL (development library, AICG_TDF_ABLATE): profiling variants -- 1 no residual loads, 2 no output stores, 4 no phase-2 weight DMA,
// 8 no phase-1 x loads, 16 no phase-1 weight DMA, 32 no phase-2 MFMAs, 64 no phase-1 MFMAs (results are garbage); 128 the next slab as one
// burst at the top of the stage (the form up to round 5; results are right)
template <int NH, int ABL = 0>
__global__ void __launch_bounds__(256) tdf_pair_kernel(TdfArgs p) {
    constexpr int H = NH * 32;
    constexpr int STAGE = tdf_stage_floats(H);
    constexpr int W2LD = H + 4;
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * TR;
    const int n1 = p.F / TK;          // phase-1 stages
    const int n2 = p.F / 32;          // phase-2 stages (32 output columns each)

    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const long row = r0 + wave * 32 + l31;                 // this lane's row (the MFMA column)
    const bool row_ok = row < p.R;
    f32x16 acc[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // Stage s + 1 is fetched while stage s computes: weight slab by DMA straight into the other buffer, x slab into four registers
    // committed at the end of the stage; the barrier that opens stage s + 1 follows this wave's vmcnt(0).
    constexpr int W1Q = TK * H / 4;                       // float4 per W1 slab
    constexpr int W2Q = tdf_w2_slab_floats(H) / 4;        // float4 per W2 slab
    constexpr int NP = (W2Q + 255) / 256;                 // DMA pieces per wave and slab (W1Q <= W2Q)
    // The next slab goes out a piece at a time between blocks of four MFMAs (level 0's block: 7.55 ms as a burst at the top of the stage,
    // 7.32 with the pieces spread, 6.75 = 137 TFLOP/s with the eight-instruction piece of tdf_dma_piece).  ABL bit 128 forces the burst.
    constexpr bool BURST = (ABL & 128) != 0;
    static_assert(NP <= 4 * NH && W1Q <= W2Q, "a stage has a block of four MFMAs per piece");
    XStage xs;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and said so: what derives from it lives in SGPRs
    const unsigned voff16 = 16u * (unsigned)tid;          // a thread's byte offset inside a DMA round of the workgroup (pieces of 1 KiB per wave)
#ifdef AICG_EMULATED
    const unsigned smem_lds = 0;
#else
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
#endif
    tdf_dma_slab(p.w1p, smem, W1Q, tid);
    tdf_load_x(p, tid, r0, 0, xs);
    tdf_commit_x(p, smem + TK * H, tid, r0, xs);
    dma_wait();
    // ---- phase 1: acc[i] = (x W1^T)^T tile i: rows h = 32 i .. 32 i + 31, column = this lane's row
    for (int st = 0; st < n1; ++st) {
        lds_barrier();   // stage st is in LDS; every wave is done with stage st - 1, whose buffer takes stage st + 1 now
        float* nbuf = smem + ((st + 1) & 1) * STAGE;
        const unsigned nbuf_lds = smem_lds + (unsigned)(((st + 1) & 1) * STAGE * 4);
        // the next stage's weight slab (the last stage: phase 2's first) goes out one piece per block of four MFMAs below
        const bool more = st + 1 < n1;
        const float* nsrc = more ? p.w1p + (long)(st + 1) * W1Q * 4 : p.w2p;
        const int nq = more ? W1Q : W2Q;
        if constexpr (BURST) tdf_dma_slab(nsrc, nbuf, nq, tid);
        if (more) {
            if constexpr ((ABL & 8) == 0) tdf_load_x(p, tid, r0, st + 1, xs);
        }
        const float4* wq = reinterpret_cast<const float4*>(smem + (st & 1) * STAGE) + half * H + l31;
        const float4* xq = reinterpret_cast<const float4*>(smem + (st & 1) * STAGE + TK * H) + half * TR + wave * 32 + l31;
#pragma unroll
        for (int g = 0; g < TK / 8; ++g) {
            const float4 b = xq[g * 2 * TR];
            float4 a = wq[g * 2 * H];
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const float4 an = wq[g * 2 * H + (i + 1 < NH ? i + 1 : i) * 32];   // next tile's quad in flight under these MFMAs
                if constexpr ((ABL & 16) == 0 && !BURST) {
                    if (g * NH + i < W1Q / 256) tdf_dma_piece(nsrc, nbuf, nbuf_lds, W1Q, tid, voff16, wave_s, g * NH + i);   // (inside both slabs)
                    else if (g * NH + i < NP && !more) tdf_dma_piece(nsrc, nbuf, nbuf_lds, W2Q, tid, voff16, wave_s, g * NH + i);
                }
                if constexpr ((ABL & 64) != 0) { acc[i][0] += a.x * b.x + an.y; a = an; continue; }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i], 0, 0, 0);
                a = an;
            }
        }
        // nothing that consumes the x loads may be scheduled above the MFMAs (without the fence hipcc hoists the parity shuffles of
        // tdf_commit_x to the top of the stage and waits for the loads -- and the DMA queued before them -- in front of the MFMAs)
        sched_fence();
        if (st + 1 < n1) tdf_commit_x(p, nbuf + TK * H, tid, r0, xs);
        dma_wait();
    }
    // ---- hand-over: bias + BatchNorm (the 32 rows of a wave share a channel: rows_per_ch % 32 == 0) + ReLU, in registers
    const int ch = (int)(((r0 + wave * 32) / p.rows_per_ch) % p.n_ch);
    {
        const float sc = p.s1 ? p.s1[ch] : 1.f, sh = p.t1 ? p.t1[ch] : 0.f;
#pragma unroll
        for (int i = 0; i < NH; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = p.b1 ? *reinterpret_cast<const float4*>(p.b1 + 32 * i + 8 * q + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (acc[i][4 * q + e] + bb[e]) * sc + sh;
                    acc[i][4 * q + e] = v > 0.f ? v : 0.f;
                }
            }
    }
    // ---- phase 2: 32 output columns per stage
    const float sc2 = p.s2 ? p.s2[ch] : 1.f, sh2 = p.t2 ? p.t2[ch] : 0.f;
    const float* xrow = p.x + row * p.F;
    float* orow = p.out + row * p.F;
    for (int fb = 0; fb < n2; ++fb) {
        const int st = n1 + fb;
        lds_barrier();
        // residual x[row][32 fb + 8 q + 4 half .. + 3]: requested before the MFMAs, consumed behind them
        float4 rx[4], bq4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rx[q] = (row_ok && (ABL & 1) == 0) ? *reinterpret_cast<const float4*>(xrow + 32 * fb + 8 * q + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
            bq4[q] = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + 32 * fb + 8 * q + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // the DMA goes out BEHIND these loads: hipcc guards the re-use of the previous stage's store-data registers with a vmcnt(0),
        // which must only meet stores issued a whole stage ago, not a DMA issued a moment ago (vmcnt retires in order)
        const float* nsrc = p.w2p + (long)(fb + 1) * W2Q * 4;
        float* const nbuf = smem + ((st + 1) & 1) * STAGE;
        const unsigned nbuf_lds = smem_lds + (unsigned)(((st + 1) & 1) * STAGE * 4);
        const bool more = fb + 1 < n2;
        if constexpr (BURST) { if (more) tdf_dma_slab(nsrc, nbuf, W2Q, tid); }
        const float4* w2q = reinterpret_cast<const float4*>(smem + (st & 1) * STAGE + l31 * W2LD) + half;   // row f = l31 of the slab
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        float4 a = w2q[0];
#pragma unroll
        for (int i = 0; i < NH; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nxt = (i * 4 + q + 1 < NH * 4) ? (i * 4 + q + 1) : (i * 4 + q);
                const float4 an = w2q[(nxt >> 2) * 8 + (nxt & 3) * 2];          // float4 index of h = 32 i' + 8 q' (+ 4 half via the base)
                if constexpr ((ABL & 4) == 0 && !BURST) {
                    if (i * 4 + q < NP && more) tdf_dma_piece(nsrc, nbuf, nbuf_lds, W2Q, tid, voff16, wave_s, i * 4 + q);
                }
                if constexpr ((ABL & 32) != 0) { o[0] += a.x * acc[i][4 * q] + an.y; a = an; continue; }
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, acc[i][4 * q], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, acc[i][4 * q + 1], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, acc[i][4 * q + 2], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, acc[i][4 * q + 3], o, 0, 0, 0);
                a = an;
            }
        dma_wait();   // the next slab's DMA (issued a whole stage ago) and this stage's loads; the stores below drain under the next stage
        if (row_ok && ((ABL & 2) == 0 || o[0] == 123.456f)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 32 * fb + 8 * q + 4 * half;
                const float4 bq = bq4[q];
                float e0 = (o[4 * q] + bq.x) * sc2 + sh2, e1 = (o[4 * q + 1] + bq.y) * sc2 + sh2;
                float e2 = (o[4 * q + 2] + bq.z) * sc2 + sh2, e3 = (o[4 * q + 3] + bq.w) * sc2 + sh2;
                e0 = (e0 > 0.f ? e0 : 0.f) + rx[q].x;
                e1 = (e1 > 0.f ? e1 : 0.f) + rx[q].y;
                e2 = (e2 > 0.f ? e2 : 0.f) + rx[q].z;
                e3 = (e3 > 0.f ? e3 : 0.f) + rx[q].w;
                *reinterpret_cast<float4*>(orow + f) = make_float4(e0, e1, e2, e3);
            }
        }
    }
}

template <int NH, int ABL = 0>
static int launch_tdf_abl(const TdfArgs& p, hipStream_t st) {
    const size_t lds = (size_t)2 * tdf_stage_floats(NH * 32) * sizeof(float);
    allow_dynamic_lds((const void*)tdf_pair_kernel<NH, ABL>, lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tdf_pair_kernel<NH, ABL>), dim3((unsigned)ldiv_up(p.R, TR)), dim3(256), lds, st, p);
    return check_launch("tdf_pair_kernel");
}

template <int NH>
static int launch_tdf(const TdfArgs& p, hipStream_t st) {
#ifdef AICG_DEV_SWITCHES
    AICG_SWITCH(tdf_ablate, "AICG_TDF_ABLATE", 0);
    if constexpr (NH == 12) {
        switch ((int)tdf_ablate) {
            case 1: return launch_tdf_abl<NH, 1>(p, st);
            case 2: return launch_tdf_abl<NH, 2>(p, st);
            case 3: return launch_tdf_abl<NH, 3>(p, st);
            case 4: return launch_tdf_abl<NH, 4>(p, st);
            case 7: return launch_tdf_abl<NH, 7>(p, st);
            case 8: return launch_tdf_abl<NH, 8>(p, st);
            case 16: return launch_tdf_abl<NH, 16>(p, st);
            case 24: return launch_tdf_abl<NH, 24>(p, st);
            case 32: return launch_tdf_abl<NH, 32>(p, st);
            case 64: return launch_tdf_abl<NH, 64>(p, st);
            case 96: return launch_tdf_abl<NH, 96>(p, st);
            case 128: return launch_tdf_abl<NH, 128>(p, st);
            case 31: return launch_tdf_abl<NH, 31>(p, st);
            default: break;
        }
    }
#endif
    return launch_tdf_abl<NH, 0>(p, st);
}

}  // namespace aicg

using namespace aicg;

extern "C" int aicg_tdf_pair_supported(int F, int H, int rows_per_ch) {
    const int nh = H / 32;
    return (H % 32 == 0 && (nh == 2 || nh == 3 || nh == 4 || nh == 6 || nh == 8 || nh == 12) && F % 32 == 0 && rows_per_ch % 32 == 0) ? 1 : 0;
}

extern "C" int aicg_tdf_pair(const float* x, const float* w1_packed, const float* b1, const float* s1, const float* t1, const float* w2_packed,
                             const float* b2, const float* s2, const float* t2, float* out, int64_t R, int F, int H, int rows_per_ch,
                             int n_ch, void* stream) {
    if (!x || !w1_packed || !w2_packed || !out) return fail(AICG_E_ARG, "aicg_tdf_pair: null pointer");
    if ((s1 != nullptr) != (t1 != nullptr) || (s2 != nullptr) != (t2 != nullptr)) return fail(AICG_E_ARG, "aicg_tdf_pair: scale without shift");
    if (!aicg_tdf_pair_supported(F, H, rows_per_ch) || n_ch < 1)
        return fail(AICG_E_SHAPE, "aicg_tdf_pair: needs H in 32 x {2,3,4,6,8,12}, F %% 32 == 0, rows_per_ch %% 32 == 0 (F %d, H %d, rows %d)", F, H, rows_per_ch);
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!al(x) || !al(w1_packed) || !al(w2_packed) || !al(out) || (b1 && !al(b1)) || (b2 && !al(b2)))
        return fail(AICG_E_ARG, "aicg_tdf_pair: operands must be 16-byte aligned");
    if (R <= 0) return AICG_OK;
    TdfArgs p{x, w1_packed, b1, s1, t1, w2_packed, b2, s2, t2, out, (long)R, F, H, rows_per_ch, n_ch};
    hipStream_t st = (hipStream_t)stream;
    switch (H / 32) {
        case 2: return launch_tdf<2>(p, st);
        case 3: return launch_tdf<3>(p, st);
        case 4: return launch_tdf<4>(p, st);
        case 6: return launch_tdf<6>(p, st);
        case 8: return launch_tdf<8>(p, st);
        default: return launch_tdf<12>(p, st);
    }
}
