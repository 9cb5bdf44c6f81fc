// 3 x 3 convolution (stride 1, dilation 1, padding 1) by the TWO-dimensional Winograd minimal-filtering form F(2 x 2, 3 x 3):
// a 2 x 2 block of outputs from a 4 x 4 block of inputs with 16 multiplications per (input channel, output channel) where the
// direct implicit GEMM spends 36 and the row-only form of conv_ws3w.h 24.
//
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A                d: input rows 2 t - 1 .. 2 t + 2, columns 2 j - 1 .. 2 j + 2
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]    A^T = [1 1 1 0; 0 1 -1 -1]
//
// (the one-dimensional matrices of conv_ws3w.h applied along both axes; every constant is a power of two, so the result differs
// from the direct form only by fp32 summation order).  The layer becomes SIXTEEN independent GEMMs (one per Winograd point
// p = 4 i + q) of K = Cin over a quarter as many columns.  U = G g G^T is folded into the packed weights at load time
// (ops.winograd2d_image); V = B^T d B is formed by the wave that consumes it, from the RAW input patch in LDS, in the issue slots
// the matrix pipe leaves free (64 VALU additions per MFMA k-step of 96 MFMAs): no producer waves, no transformed planes in LDS --
// the ablation of the row-only kernel priced its producers' VALU issue and returning loads at 0.7 of 3.67 ms.
//
// Geometry.  hipcc keeps MFMA accumulators in the 256 AGPRs of a wave, so a wave owns 16 points x 48 output channels (three 16-row
// blocks: 48 is the channel granularity of every MDX-Net level) x 16 Winograd tiles = 192 accumulator registers of
// v_mfma_f32_16x16x4_f32.  A workgroup of NW waves owns one M unit of 48 output channels x (NW output rows x 64 output columns) of one
// image; wave w owns row pair w >> 1 and tile block w & 1 (column pairs 16 (w & 1) .. + 15).  NW = 8 (two waves per SIMD: one
// wave's issue bubbles are the other's MFMA time; 192 AGPRs + 64 VGPRs each) or 4 (one per SIMD, any number of VGPRs, half the tile).
// K runs in chunks of 8 input channels = 2 MFMA k-steps (lane group ks = lane >> 4 contracts channel 4 s + ks in step s).
// One LDS stage per chunk, THREE buffers, filled by LDS DMA (buffer_load_dwordx4 ... lds: 1 KiB per wave-instruction, no staging
// registers, zero padding by the buffer range check) that the same waves issue two stages ahead:
//     weights  [s][p][ks][m = 0..47]  floats: 24 KiB, a contiguous slab of the packed image
//     patch    [channel c = 0..7][row 0..NW+1][18 quads] (+ pad quads: the plane stride is 8 mod 16 quads, which puts the four ks groups
//              of an 8-byte read on disjoint banks): input rows h0 - 1 .. h0 + NW, columns w0 - 4 .. w0 + 67, 16-byte aligned in HBM.
// The k-steps form one software pipeline across chunk, tile and item boundaries: while the 48 MFMAs of k-step u run (16 points x 3
// row blocks, one A fragment each), the wave reads the 4 x 4 raw patch of k-step u + 1 (a lane = tile column l15, channel ks; three
// aligned 8-byte reads per patch row), forms its 16 V values a few additions per point, and prefetches the next point's fragments.
// Stage g + 1 is complete in LDS when stage g starts (its DMA was waited for before the barrier), which is what lets k-step (g, 1)
// read ahead into it; the DMA of stage g + 2 goes into the third buffer.
// Epilogue: A^T M A in registers -- the 16 accumulators of an output never meet another lane --, bias, activation, float2 stores (the
// first k-step of an item starts its accumulators from the inline constant 0 instead of adding to zeroed registers).
// Persistent walk over (tile, M unit) items, M unit fastest, XCD x owning a contiguous eighth of the list: the workgroups that
// share an input patch run side by side on one L2.
#pragma once
#include "conv_kernels.h"

namespace aicg {

typedef float w2d_f32x4 __attribute__((ext_vector_type(4)));

static constexpr int kW2dM = 48;                                  // output channels per M unit
static constexpr int kW2dCols = 64;                               // output columns of a workgroup's tile (its rows: NW)
static constexpr int kW2dPQuads = (kW2dCols + 8) / 4;             // 18: patch columns w0 - 4 .. w0 + 67
static constexpr int kW2dWFloats = 2 * 16 * 4 * kW2dM;            // 6144 floats of weights per chunk
static constexpr int kW2dWPieces = kW2dWFloats / 256;             // 24 DMA pieces (1 KiB) of weights per chunk
// quads per channel plane of the patch: (NW + 2) rows of 18, rounded up to 8 mod 16
__host__ __device__ constexpr int w2d_plane_quads(int nw) { return ((nw + 2) * kW2dPQuads + 7) / 16 * 16 + 8; }
__host__ __device__ constexpr int w2d_stage_floats(int nw) { return kW2dWFloats + 8 * w2d_plane_quads(nw) * 4; }

// One LDS-DMA piece: lane l fetches 16 bytes at buffer offset voff (+ the wave-uniform soff) and the hardware writes them to
// lds_wave_base + 16 l.  Offsets >= the resource's num_records (kBufOob) deposit zeros -- the convolution's padding.
__device__ __forceinline__ void w2d_dma16(const BufRsrc& r, unsigned voff, unsigned soff, float* lds_wave_base, int lane) {
#ifdef AICG_EMULATED
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned long)voff + 16 <= r.num_records) __builtin_memcpy(&v, r.base + voff + soff, 16);
    *reinterpret_cast<float4*>(lds_wave_base + 4 * lane) = v;
#else
    (void)lane;
    // inline asm on purpose (tdf_pair.hip): with the builtin hipcc guards every following ds_read with a vmcnt(0); the stage-end
    // w2d_dma_wait() orders it by hand.  M0 = LDS base of the piece, restored.
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    buf_i32x4 rs;   // wave-uniform by construction; say so (a resource the compiler cannot prove uniform lands in VGPRs)
    rs.x = __builtin_amdgcn_readfirstlane(r.d.x); rs.y = __builtin_amdgcn_readfirstlane(r.d.y);
    rs.z = __builtin_amdgcn_readfirstlane(r.d.z); rs.w = __builtin_amdgcn_readfirstlane(r.d.w);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(lds), "s"(so)
                 : "memory");
#endif
}
// A RUN of N <= 4 pieces of one resource whose LDS destinations are consecutive KiB (piece e lands at lds_base + 1 KiB x e), issued as ONE
// statement with one M0 save / restore (N calls of w2d_dma16 are 9 N instructions + a generic-to-LDS pointer cast with its null check
// each; profiles/NOTES.md R6: the probe's L2-resident burst of 48 pieces lands at 44 B per clock and CU, while conv_w2d's waves stood
// ~3 000 cycles per stage in ~150 issue slots of address arithmetic and three exposed LDS reads).
//   SAME: a contiguous slab (the weights) -- every piece uses the per-lane offset v[0] and the wave-uniform `soff`; the instruction's
//         12-bit immediate advances the memory AND the LDS address by 1 KiB x e (LDS_ADDR = M0 base + inst_offset + 16 x lane): N + 4
//         instructions;
//   else: piece e has its own per-lane offsets v[e] (the patch; kBufOob = padding) and M0 itself is advanced: 3 N + 2 instructions.
template <int N, bool SAME>
__device__ __forceinline__ void w2d_dma_run(const BufRsrc& r, const unsigned (&v)[4], unsigned soff, float* lds_base, unsigned lds_addr, int lane) {
    static_assert(N >= 1 && N <= 4, "the immediate offset has 12 bits");
#ifdef AICG_EMULATED
    (void)lds_addr;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const unsigned voff = SAME ? v[0] + 1024u * e : v[e];      // the hardware range-checks voff + the immediate (not the scalar offset)
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned long)voff + 16 <= r.num_records) __builtin_memcpy(&q, r.base + voff + soff, 16);
        *reinterpret_cast<float4*>(lds_base + 256 * e + 4 * lane) = q;
    }
#else
    (void)lane; (void)lds_base;
    buf_i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane(r.d.x); rs.y = __builtin_amdgcn_readfirstlane(r.d.y);
    rs.z = __builtin_amdgcn_readfirstlane(r.d.z); rs.w = __builtin_amdgcn_readfirstlane(r.d.w);
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr);
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    unsigned keep;
#define W2D_HEAD "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
#define W2D_TAIL "s_mov_b32 m0, %0"
#define W2D_BUMP "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
    if constexpr (SAME) {
        if constexpr (N == 1)
            asm volatile(W2D_HEAD "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t" W2D_TAIL
                         : "=&s"(keep) : "s"(rs), "s"(lds), "s"(so), "v"(v[0]) : "memory");
        else if constexpr (N == 2)
            asm volatile(W2D_HEAD "buffer_load_dwordx4 %4, %1, %3 offen lds\n\tbuffer_load_dwordx4 %4, %1, %3 offen offset:1024 lds\n\t" W2D_TAIL
                         : "=&s"(keep) : "s"(rs), "s"(lds), "s"(so), "v"(v[0]) : "memory");
        else if constexpr (N == 3)
            asm volatile(W2D_HEAD "buffer_load_dwordx4 %4, %1, %3 offen lds\n\tbuffer_load_dwordx4 %4, %1, %3 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %4, %1, %3 offen offset:2048 lds\n\t" W2D_TAIL
                         : "=&s"(keep) : "s"(rs), "s"(lds), "s"(so), "v"(v[0]) : "memory");
        else
            asm volatile(W2D_HEAD "buffer_load_dwordx4 %4, %1, %3 offen lds\n\tbuffer_load_dwordx4 %4, %1, %3 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %4, %1, %3 offen offset:2048 lds\n\tbuffer_load_dwordx4 %4, %1, %3 offen offset:3072 lds\n\t" W2D_TAIL
                         : "=&s"(keep) : "s"(rs), "s"(lds), "s"(so), "v"(v[0]) : "memory");
    } else {
        if constexpr (N == 1)
            asm volatile(W2D_HEAD "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t" W2D_TAIL
                         : "=&s"(keep) : "s"(rs), "s"(lds), "s"(so), "v"(v[0]) : "memory");
        else if constexpr (N == 2)
            asm volatile(W2D_HEAD "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t" W2D_BUMP "buffer_load_dwordx4 %5, %1, %3 offen lds\n\t" W2D_TAIL
                         : "=&s"(keep) : "s"(rs), "s"(lds), "s"(so), "v"(v[0]), "v"(v[1]) : "memory", "scc");
        else if constexpr (N == 3)
            asm volatile(W2D_HEAD "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t" W2D_BUMP "buffer_load_dwordx4 %5, %1, %3 offen lds\n\t"
                         W2D_BUMP "buffer_load_dwordx4 %6, %1, %3 offen lds\n\t" W2D_TAIL
                         : "=&s"(keep) : "s"(rs), "s"(lds), "s"(so), "v"(v[0]), "v"(v[1]), "v"(v[2]) : "memory", "scc");
        else
            asm volatile(W2D_HEAD "buffer_load_dwordx4 %4, %1, %3 offen lds\n\t" W2D_BUMP "buffer_load_dwordx4 %5, %1, %3 offen lds\n\t"
                         W2D_BUMP "buffer_load_dwordx4 %6, %1, %3 offen lds\n\t" W2D_BUMP "buffer_load_dwordx4 %7, %1, %3 offen lds\n\t" W2D_TAIL
                         : "=&s"(keep) : "s"(rs), "s"(lds), "s"(so), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory", "scc");
    }
#undef W2D_HEAD
#undef W2D_TAIL
#undef W2D_BUMP
#endif
}
template <bool SKIP = false>
__device__ __forceinline__ void w2d_dma_wait() {
    if constexpr (SKIP) return;                      // (ABL 32768: what the stage-end wait costs; races, wrong results)
#ifndef AICG_EMULATED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
__device__ __forceinline__ void w2d_fence() {
#ifndef AICG_EMULATED
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// One MFMA of the k-step pipeline, accumulating IN PLACE.  TIED: inline asm whose "+v" operand pins the destination to the addend.  With
// the builtin hipcc is free to give v_mfma a destination other than its addend, and in the straight-line k-steps around the steady loop
// (an item's first stage behind its zero-started k-step, its whole last stage) it does -- it uses the MFMA to move accumulator quads into
// the positions the next region wants -- at the price of both copies being live at once: the 192 accumulators + 64 registers of the
// eight-wave form then do not fit and accumulator quads go through scratch memory inside the MFMA stream (s_nop 9 + scratch_store per
// quad, a vmcnt(0) in front of the reload that also waits for the epilogue's stores).  The hazard recogniser does not see inside inline
// asm: the only non-MFMA readers of an accumulator are the epilogue's VALU instructions, behind w2d_mfma_settle().
template <bool TIED>
__device__ __forceinline__ void w2d_mfma_acc(w2d_f32x4& c, float a, float b) {
#ifdef AICG_EMULATED
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#else
    if constexpr (TIED) asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}
template <bool TIED>
__device__ __forceinline__ void w2d_mfma_zero(w2d_f32x4& c, float a, float b) {   // an item's first k-step: the addend is the inline constant 0
#ifdef AICG_EMULATED
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, w2d_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#else
    if constexpr (TIED) asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    else c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, w2d_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#endif
}
// an 8-pass MFMA's result may be read by a VALU / VMEM instruction 11 wait states after it issued (hipcc inserts them for the builtin)
__device__ __forceinline__ void w2d_mfma_settle() {
#ifndef AICG_EMULATED
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#endif
}
template <class T>
__device__ inline T w2d_opaque(T v) {   // a per-item copy of a uniform value the optimiser cannot hoist out of the walk
#ifndef AICG_EMULATED
    asm volatile("" : "+s"(v));
#endif
    return v;
}

// PF 1 / 3: points a fragment is fetched ahead (dword fragments: weights [s][p][ks][m]); PF == 0: QUAD fragments -- weights
// [s][p / 4][ks][m][p % 4], one ds_read_b128 per (point group, row block) feeding four MFMAs, reloaded in place behind them; PF == 2: PAIR
// fragments -- [s][p / 2][ks][m][p % 2], one ds_read_b64 per two MFMAs.
// ABL: profiling variants, instantiated in the dev library only (tools/kbench_w2d_ablate.py): 1 no DMA, 2 no fragment reads, 4 no patch
// reads / transform, 8 no MFMAs, 16 no epilogue, 32 no stage barriers, 64 clocks of workgroup 0 into y[0..1], 128 epilogue without its
// stores, 8192 place() at the top of the stage that needs it, 512 the round-4 / 5 stage burst (one w2d_dma16 per piece, pieces w, w + NW, ... of the patch) for A/B against the runs of
// w2d_dma_run.  Compile-time: a run-time switch in the k-step loop costs the 8-wave form its register budget.
// KS: k-steps per LDS stage.  2 (routed): stages of 8 input channels.  1 (development builds, aicg_conv_desc.wino 16): stages of 4 -- half
// the weights slab (12 KiB) and four patch planes per stage, three stage buffers in < 64 KiB, so that TWO four-wave workgroups fit a CU
// (each SIMD holds one wave of each) and what one workgroup cannot hide -- its barriers, its bursts, an item's epilogue and pipeline
// restart -- could run under the other's MFMAs.  Compiles to 256 registers without a spill, bit-identical output -- and runs level for
// level as fast as the eight-wave form, 2.42 / 2.16 / 1.18 / 0.52 / 0.22 ms against 2.42 / 2.15 / 1.17 / 0.52 / 0.22, with or without a start
// skew of half an item between the two workgroups of a CU (profiles/r06_kbench_w2d_two_workgroups*.txt): exposed synchronisation is not
// what the kernel waits for.
template <int NW, int PF, int ABL = 0, int KS = 2>
__global__ void __launch_bounds__(64 * NW, KS == 1 ? 2 : 1) conv_w2d_kernel(ConvArgs p) {
    constexpr int dbg = ABL;
    constexpr bool OLD_DMA = (ABL & 512) != 0;
    constexpr bool PK_EPI = (ABL & 65536) == 0;       // the epilogue's additions packed over channel PAIRS (ABL 65536: one channel at a time, for A/B)
    constexpr bool TIED = (ABL & 16384) == 0;         // MFMAs through inline asm with the destination tied to the addend (w2d_mfma_acc); ABL 16384: the builtin, for A/B
    constexpr int BUFS = 3;                          // stage g computes, stage g + 1 has landed (k-step (g, 1) reads ahead into it), stage g + 2 is being filled
    static_assert(PF == 0 || PF == 1 || PF == 2 || PF == 3, "PF 1 / 3: a fragment ring of PF + 1 slots (16 points are a whole number of turns); 0 / 2: groups");
    // PF == 0 / 2: GROUP fragments -- the weights of G = 4 / 2 consecutive points side by side in LDS ([s][p / G][ks][m][p % G]): one
    // ds_read_b128 / ds_read_b64 per (point group, row block) feeds G MFMAs and is reloaded in place right behind them.  Every LDS read
    // whose result the matrix pipe consumes costs the pipe a bubble whatever its width (profiles/NOTES.md 2.2: 256 against 192 cycles per
    // k-step with dword fragments, 204 with one b128 per four): pairs halve the 96 reads of a stage in the SAME six registers the
    // dword ring holds (quads need twelve: 77 spills in the eight-wave form).
    constexpr bool AQ = PF == 0 || PF == 2;
    constexpr int G = PF == 0 ? 4 : 2;                // points per fragment group (AQ)
    constexpr int NT = 64 * NW;
    constexpr int PROWS = NW + 2;                                 // patch rows
    constexpr int PLANEQ = w2d_plane_quads(NW);
    constexpr int CHS = 4 * KS;                                   // input channels per stage
    constexpr int WFLOATS = kW2dWFloats * KS / 2;                 // floats of weights per stage
    constexpr int WPIECES = WFLOATS / 256;
    constexpr int PPIECES = (CHS * PLANEQ + 63) / 64;             // DMA pieces of patch per stage (23 / 15; KS 1: the last one may be ragged)
    constexpr int STAGE = WFLOATS + PPIECES * 256;
    static_assert(WPIECES % NW == 0 && (KS == 1 || CHS * PLANEQ % 64 == 0), "pieces are whole");
    static_assert(KS == 2 || KS == 1, "one or two k-steps per stage");
    HIP_DYNAMIC_SHARED(float4, smem4)
    float* const smem = reinterpret_cast<float*>(smem4);
    float* const bias_s = smem + BUFS * STAGE;       // Cout: the layer's bias
#ifdef AICG_EMULATED
    const unsigned smem_lds = 0;
#else
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and said so: everything derived from it lives in SGPRs
    // Per-lane constants are RE-DERIVED from the thread id wherever they are used instead of being kept: besides its 192 accumulators a
    // wave of the eight-wave form has 64 registers, and a value the compiler spills comes back through scratch memory -- a VMEM load
    // whose s_waitcnt vmcnt(0) also waits for every DMA piece issued before it.
    auto lane_now = [&]() __attribute__((always_inline)) {
        int t = threadIdx.x & 63;
#ifndef AICG_EMULATED
        asm volatile("" : "+v"(t));
#endif
        return t;
    };
    const int nmu = p.Mpad;                          // M units (launch_conv_w2d stores Cout / 48 here)
    const int nchunk = p.nchunk;
    // persistent walk: item = tile * nmu + mu; XCD x owns a contiguous eighth of the item list, its workgroups take it `slots` at a time
    const int ntiles = p.N * p.tiles_h * p.tiles_w;
    const int nitems = ntiles * nmu;
    const int slots = gridDim.x >> 3, per_xcd = (nitems + 7) >> 3;
    const int first = (blockIdx.x & 7) * per_xcd, slot = blockIdx.x >> 3;
    const int mine = nitems - first < per_xcd ? nitems - first : per_xcd;
    const int my_items = slot < mine ? (mine - slot + slots - 1) / slots : 0;
    if (my_items <= 0) return;

    for (int i = tid; i < nmu * kW2dM; i += NT) bias_s[i] = p.bias ? p.bias[i] : 0.f;

    // ---- DMA plan of this wave: weight pieces w WPW .. + WPW - 1, patch pieces w, w + NW, ...  What a lane fetches of a patch piece --
    // quad (channel c, patch row, quad column) -- is the same for every tile: decoded once into LDS (c << 16 | row << 8 | column, or ~0
    // for an unused slot); per item the byte offsets are rebuilt from it, again into LDS ([piece][thread] dwords behind the bias).
    constexpr int WPW = WPIECES / NW;                // weight pieces per wave (3 / 6; KS 1, four waves: 3)
    constexpr int NPP = (PPIECES + NW - 1) / NW;     // patch pieces per wave (3 / 4)
    unsigned* const tab_s = reinterpret_cast<unsigned*>(bias_s + ((nmu * kW2dM + 3) & ~3));   // [2][NPP][NT]: decode, then offsets
#pragma unroll
    for (int e = 0; e < NPP; ++e) {
        const int piece = OLD_DMA ? wave + NW * e : wave * NPP + e;   // a wave's pieces are consecutive KiB of the stage: one run (w2d_dma_run)
        const int Q = piece * 64 + (tid & 63);
        const int c = Q / PLANEQ, rem = Q - c * PLANEQ;
        const int row = rem / kW2dPQuads, qd = rem - row * kW2dPQuads;
        tab_s[e * NT + tid] = (piece < PPIECES && c < CHS && rem < PROWS * kW2dPQuads) ? (unsigned)(c << 16 | row << 8 | qd) : 0xffffffffu;
    }
    const float* xg = p.x;
    int li = 0, lc = 0, lmu = 0;                     // DMA cursor: item of this workgroup, chunk; M unit of that item
    auto place = [&](int item) __attribute__((always_inline)) {
        const int tile = item / nmu;
        lmu = item - tile * nmu;
        const int tw_i = tile % p.tiles_w, th_i = (tile / p.tiles_w) % p.tiles_h, n = tile / (p.tiles_w * p.tiles_h);
        const int w0 = tw_i * kW2dCols, h0 = th_i * NW;
        xg = p.x + (long)n * p.x_sn;
        unsigned* tl = tab_s + wave * 64 + lane_now();
#pragma unroll
        for (int e = 0; e < NPP; ++e) {
            const unsigned dcd = tl[e * NT];
            const int c = (int)(dcd >> 16), row = (int)((dcd >> 8) & 255u), qd = (int)(dcd & 255u);
            const int hin = h0 - 1 + row, win = w0 - 4 + 4 * qd;
            const bool ok = dcd != 0xffffffffu && hin >= 0 && hin < p.H && win >= 0 && win + 4 <= p.W;
            // ABL 1024: every patch piece reads the same L2-resident KiB (wrong results): what the HBM side of the burst costs
            if constexpr ((dbg & 1024) != 0) tl[(NPP + e) * NT] = 16u * (unsigned)lane_now();
            else tl[(NPP + e) * NT] = ok ? 4u * (unsigned)(c * (int)p.x_sc + hin * (int)p.x_sh + win) : kBufOob;
        }
    };
    // place() ahead of time: a stage whose burst will open a new item places it BEHIND its first k-step of the stage before -- ~80 scalar
    // and ~100 vector instructions (divisions, range checks) that stood at the top of a stage, where all eight waves run them with the
    // matrix pipe idle; behind a k-step the SIMD's other wave covers them
    bool placed = false;
    auto place_ahead = [&]() __attribute__((always_inline)) {
        if constexpr (OLD_DMA || (ABL & 8192) != 0) return;      // (ABL 8192: at the top of the stage, as in rounds 4-5)
        if (li < my_items && lc == 0 && !placed) { place(first + slot + li * slots); placed = true; }
    };
    auto issue = [&](float* buf) __attribute__((always_inline)) {   // the next stage of the walk = (item li, chunk lc) into `buf`: one burst
        if (li >= my_items) return;
        if constexpr ((dbg & 1) != 0) { if (++lc == nchunk) { lc = 0; ++li; } return; }
        if (lc == 0 && !placed) place(first + slot + li * slots);
        placed = false;
        const int lane = lane_now();
        // (the image is [M unit][8-channel chunk][k-step] slabs: with KS == 1 a stage is one k-step's half slab, `nchunk` counts those)
        const long wbase = ((long)lmu * nchunk + lc) * WFLOATS;
        const BufRsrc wb = make_buf(p.w3 + wbase, (unsigned)(WFLOATS * 4));
        const long left = (long)(p.Cin_g - lc * CHS) * p.x_sc * 4;   // bytes up to the end of the image's channels: absent channels read 0
        const BufRsrc xb = make_buf(xg + (long)lc * CHS * p.x_sc, (unsigned)lmin(left, 0x7fffffffL));
        if constexpr (OLD_DMA) {
#pragma unroll
            for (int e = 0; e < WPW; ++e) {
                const int piece = wave * WPW + e;
                w2d_dma16(wb, 16u * (unsigned)lane, 1024u * (unsigned)piece, buf + piece * 256, lane);
            }
            const unsigned* tl = tab_s + NPP * NT + wave * 64 + lane;
#pragma unroll
            for (int e = 0; e < NPP; ++e) {
                const int piece = wave + NW * e;
                if (piece < PPIECES) w2d_dma16(xb, tl[e * NT], 0u, buf + WFLOATS + piece * 256, lane);
            }
        } else {
            // the stage buffer's LDS byte address as scalar arithmetic on the workgroup's LDS base (no generic-to-LDS cast per piece)
            const unsigned buf_lds = smem_lds + (unsigned)(buf - smem) * 4u;
            // the patch offsets of this wave's pieces: read FIRST, they land under the weight pieces
            const unsigned* tl = tab_s + NPP * NT + wave * 64 + lane;
            unsigned po[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < NPP; ++e) po[e] = tl[e * NT];
            const unsigned wv[4] = {16u * (unsigned)lane, 0u, 0u, 0u};
            static_assert(WPW == 3 || WPW == 6, "weight pieces per wave");
            w2d_dma_run<3, true>(wb, wv, 1024u * (unsigned)(wave * WPW), buf + wave * WPW * 256, buf_lds + 1024u * (unsigned)(wave * WPW), lane);
            if constexpr (WPW == 6)
                w2d_dma_run<3, true>(wb, wv, 1024u * (unsigned)(wave * WPW + 3), buf + (wave * WPW + 3) * 256,
                                     buf_lds + 1024u * (unsigned)(wave * WPW + 3), lane);
            // this wave's patch pieces wave NPP .. + NPP - 1 (the last wave may own fewer: PPIECES is not a multiple of NW)
            float* pdst = buf + WFLOATS + wave * NPP * 256;
            const unsigned pdst_lds = buf_lds + (unsigned)(WFLOATS * 4) + 1024u * (unsigned)(wave * NPP);
            const int mine_p = PPIECES - wave * NPP;          // >= NPP except on the last wave
            if constexpr ((dbg & 2048) != 0) { (void)pdst; (void)pdst_lds; (void)mine_p; }   // ABL 2048: weights only (wrong results)
            else if (mine_p >= NPP) w2d_dma_run<NPP, false>(xb, po, 0u, pdst, pdst_lds, lane);
            else {
                constexpr int LASTN = PPIECES - (NW - 1) * NPP;   // 23 - 21 = 2 (eight waves), 15 - 12 = 3 (four)
                static_assert(LASTN >= 1 && LASTN <= NPP, "the last wave owns at least one piece");
                w2d_dma_run<LASTN, false>(xb, po, 0u, pdst, pdst_lds, lane);
            }
        }
        if (++lc == nchunk) { lc = 0; ++li; }
    };

    // ---- the k-step pipeline's registers
    w2d_f32x4 acc[16][3];
    float V[16];                                     // B operands of the running k-step, per point
    float2 raw[4][3];                                // the next k-step's patch as loaded: [patch row][aligned column pair]
    float N[4][4];                                   // ... on its way to V: R = B^T d, then V = R B in place
    float ring[PF + 1][3];                           // A fragments: slot pt & PF holds point pt's, fetched PF points ahead
    float4 aq[3];                                    // AQ, G == 4: the running point group's fragments per row block (x .. w = points 4 pg .. + 3)
    float2 ap[3];                                    // AQ, G == 2: ... (x, y = points 2 pg, 2 pg + 1)
    // lane-relative LDS offsets: patch (in float2 units: 8-byte reads) = plane of channel ks, patch row 2 (w >> 1), column pair
    // 16 (w & 1) + l15 -- the aligned pair that starts at input column 2 j - 2; weights (floats) = row l15 of lane group ks
    int p_lane2 = 0, w_lane = 0, l15 = 0, ks = 0;
    auto lane_offsets = [&]() __attribute__((always_inline)) {                      // (re-derived at the top of every stage, see lane_now)
        const int lane = lane_now();
        l15 = lane & 15;
        ks = lane >> 4;
        p_lane2 = ks * PLANEQ * 2 + (2 * (wave >> 1)) * kW2dPQuads * 2 + 1 + 16 * (wave & 1) + l15;
        w_lane = ks * kW2dM + l15;
    };
    lane_offsets();
    auto load_raw = [&](const float* stage, int s, int r0, int r1) __attribute__((always_inline)) {   // patch rows r0 .. r1 - 1 of k-step s of the stage at `stage`
        // (stage buffers are 16-byte aligned and every term of the index is a whole float2: 8-byte reads)
        const float2* pl = reinterpret_cast<const float2*>(__builtin_assume_aligned(stage + WFLOATS, 16)) + p_lane2 + s * 4 * PLANEQ * 2;
#pragma unroll
        for (int r = r0; r < r1; ++r)
#pragma unroll
            for (int h = 0; h < 3; ++h) raw[r][h] = pl[r * kW2dPQuads * 2 + h];
    };
    auto rows_to_R = [&](int q0, int q1) __attribute__((always_inline)) {           // columns q0 .. q1 - 1 of B^T d from the loaded patch
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            // input column 2 j - 1 + q = element q + 1 of the six loaded values
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = ((q + 1) & 1) ? raw[r][(q + 1) >> 1].y : raw[r][(q + 1) >> 1].x;
            N[0][q] = d[0] - d[2];
            N[1][q] = d[1] + d[2];
            N[2][q] = d[2] - d[1];
            N[3][q] = d[1] - d[3];
        }
    };
    auto R_to_V = [&](int i0, int i1) __attribute__((always_inline)) {              // rows i0 .. i1 - 1 of (B^T d) B, in place
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const float r0 = N[i][0], r1 = N[i][1], r2 = N[i][2], r3 = N[i][3];
            N[i][0] = r0 - r2;
            N[i][1] = r1 + r2;
            N[i][2] = r2 - r1;
            N[i][3] = r1 - r3;
        }
    };
    auto take_V = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) V[4 * i + q] = N[i][q];
#ifndef AICG_EMULATED
        // TIED: the MFMAs that read V are inline asm, which the hazard recogniser does not see -- a VALU result needs two wait states
        // before an MFMA reads it (hipcc puts an s_nop 1 there for the builtin).  One s_nop per k-step with all sixteen V as operands:
        // every V is written in front of it, every MFMA of the next k-step reads the values behind it.
        if constexpr (TIED)
            asm volatile("s_nop 1" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]), "+v"(V[8]),
                         "+v"(V[9]), "+v"(V[10]), "+v"(V[11]), "+v"(V[12]), "+v"(V[13]), "+v"(V[14]), "+v"(V[15]));
#endif
    };
    auto fetch_q = [&](const float* stage, int s, int pg, int rb) __attribute__((always_inline)) {      // AQ: row block rb's fragments of point group pg of k-step s
        if constexpr ((dbg & 2) != 0) return;
        if constexpr (G == 4) aq[rb] = reinterpret_cast<const float4*>(__builtin_assume_aligned(stage, 16))[(s * 4 + pg) * 4 * kW2dM + ks * kW2dM + rb * 16 + l15];
        else ap[rb] = reinterpret_cast<const float2*>(__builtin_assume_aligned(stage, 16))[(s * 8 + pg) * 4 * kW2dM + ks * kW2dM + rb * 16 + l15];
    };
    auto fetch_a = [&](const float* stage, int s, int pt, int slot_) __attribute__((always_inline)) {   // fragments of point pt of k-step s of the stage at `stage`
        if constexpr ((dbg & 2) != 0) return;
        const float* wq = stage + (s * 16 + pt) * 4 * kW2dM + w_lane;
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) ring[slot_][rb] = wq[rb * 16];
    };
    // One k-step: 16 points x 3 MFMAs on (cur, s) with the operands in V, while the next k-step (stage nxt, k-step index sn) is read,
    // transformed and its first fragments fetched.  Every point is two scheduling regions: LDS reads (the fragments of point pt + PF,
    // the raw patch under points 8 and 9), then the point's MFMAs with the VALU of the transform between them -- left to itself hipcc
    // sinks each fragment read to just in front of the MFMA that consumes it.  The preparation is placed late (the last additions
    // under point 15) so that its registers are the ones the V values of the points already done leave free: two waves per SIMD leave
    // a wave 64 registers besides its 192 accumulators.  FIRST: the item's first k-step starts the accumulators.
    auto kstep = [&](auto first_tag, auto prep_tag, const float* cur, int s, const float* nxt, int sn) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool PREP = decltype(prep_tag)::value;     // read and transform the next k-step's patch, fetch its first fragments
        if constexpr (AQ) {
            // 3 x 16 / G steps (point group pg, row block rb) of G MFMAs; the step's fragment group is reloaded in place right behind
            // them with the same row block's group of the next point group (next k-step after the last): two steps = 2 G MFMAs of this
            // wave (and as many of the SIMD's other wave) to land.  The preparation of the next k-step sits where it sat in the quad
            // form: patch rows read behind steps at 1/2 and 7/12 of the k-step, transformed under the last third.
            constexpr int NG = 16 / G, NST = 3 * NG, SC = NST / 12;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int pg = st / 3, rb = st - 3 * pg;
                w2d_fence();
                if constexpr ((dbg & 4) == 0 && PREP) {
                    if (st == 8 * SC) rows_to_R(0, 2);
                    else if (st == 9 * SC) rows_to_R(2, 4);
                    else if (st == 10 * SC) R_to_V(0, 2);
                    else if (st == 11 * SC) R_to_V(2, 4);
                }
#pragma unroll
                for (int p4 = 0; p4 < G; ++p4) {
                    const int pt = G * pg + p4;
                    float av;
                    if constexpr (G == 4) av = p4 == 0 ? aq[rb].x : p4 == 1 ? aq[rb].y : p4 == 2 ? aq[rb].z : aq[rb].w;
                    else av = p4 == 0 ? ap[rb].x : ap[rb].y;
                    if constexpr ((dbg & 8) != 0) { if constexpr (FIRST) acc[pt][rb] = w2d_f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
                    if constexpr (FIRST) w2d_mfma_zero<TIED>(acc[pt][rb], av, V[pt]);
                    else w2d_mfma_acc<TIED>(acc[pt][rb], av, V[pt]);
                }
                w2d_fence();
                if (pg < NG - 1) fetch_q(cur, s, pg + 1, rb);
                else if constexpr (PREP) fetch_q(nxt, sn, 0, rb);
                if constexpr ((dbg & 4) == 0 && PREP) {
                    if (st == 6 * SC) load_raw(nxt, sn, 0, 2);
                    else if (st == 7 * SC) load_raw(nxt, sn, 2, 4);
                }
            }
            if constexpr (PREP) take_V();
            return;
        }
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) {
            if (pt + PF < 16) fetch_a(cur, s, pt + PF, (pt + PF) & PF);
            else if constexpr (PREP) fetch_a(nxt, sn, pt + PF - 16, (pt + PF) & PF);
            if constexpr ((dbg & 4) == 0 && PREP) {
                if (pt == 8 || pt == 9) load_raw(nxt, sn, 2 * (pt - 8), 2 * (pt - 8) + 2);
            }
            w2d_fence();
            if constexpr ((dbg & 4) == 0 && PREP) {
                if (pt >= 10 && pt < 14) rows_to_R(pt - 10, pt - 9);
                else if (pt == 14) R_to_V(0, 2);
                else if (pt == 15) R_to_V(2, 4);
            }
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                if constexpr ((dbg & 8) != 0) { if constexpr (FIRST) acc[pt][rb] = w2d_f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
                if constexpr (FIRST) w2d_mfma_zero<TIED>(acc[pt][rb], ring[pt & PF][rb], V[pt]);
                else w2d_mfma_acc<TIED>(acc[pt][rb], ring[pt & PF][rb], V[pt]);
            }
            w2d_fence();
        }
        if constexpr (PREP) take_V();
    };

    // ---- prologue: stages 0 and 1 in flight (BUFS == 2: stage 0), the first k-step's operands by hand
    float* b_cur = smem;                             // stage g, g + 1 and the one being filled (g + 2): rotate per stage
    float* b_nxt = smem + STAGE;
    float* b_fill = smem + (BUFS - 1) * STAGE;
    auto open_kstep = [&](const float* stage_) __attribute__((always_inline)) {     // operands of k-step 0 of the stage at `stage_`, without anything to hide behind
        load_raw(stage_, 0, 0, 4);
        rows_to_R(0, 4);
        R_to_V(0, 4);
        take_V();
#pragma unroll
        for (int pt = 0; pt < PF; ++pt) fetch_a(stage_, 0, pt, pt);
        if constexpr (AQ) {
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) fetch_q(stage_, 0, 0, rb);
        }
    };
    issue(b_cur);
    issue(b_nxt);
    w2d_dma_wait();
    lds_barrier();                                   // the first stage(s) and the bias are in LDS
#ifndef AICG_EMULATED
    long long clk0 = 0, wall0 = 0;
    if constexpr ((dbg & (64 | 256)) != 0) { clk0 = clock64(); wall0 = wall_clock64(); }
#endif
    // ABL bit 256: where wave 0 of workgroup 0 spends its cycles, per PHASE, summed in registers and written once at the end (the round-5
    // form stored a time stamp per event: VMEM stores in the same in-order queue as the DMA pieces, which is what its "3 000-cycle
    // burst" measured).  Phases: 0 barrier, 1 burst + lane offsets (+ open_kstep in an item's first stage), 2 k-step 0 (+ a first stage's
    // burst / a steady stage's placement behind it), 3 k-step 1, 4 DMA wait, 5 epilogue
    long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#ifndef AICG_EMULATED
    if constexpr ((dbg & 256) != 0) tlast = clock64();
#endif
    auto stamp = [&](auto phase_tag) __attribute__((always_inline)) {
#ifndef AICG_EMULATED
        if constexpr ((dbg & 256) != 0) {
            const long long t = clock64();
            tph[decltype(phase_tag)::value] += t - tlast;
            tlast = t;
        }
#endif
    };
    // one stage: barrier (the stage(s) ahead have landed -- every wave waited for its own DMA before arriving -- and the buffer to fill
    // is free), DMA of the stage BUFS - 1 ahead, the two k-steps, wait for this wave's DMA, rotate the buffers
    // The pipeline runs WITHIN an item: its last k-step prepares nothing, so that no operand of the k-step pipeline is live across the
    // epilogue (whose temporaries would push it into scratch memory), and the next item opens with open_kstep.
    bool first_stage = true;
    auto stage = [&](auto first_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        if (!first_stage && (dbg & 32) == 0) lds_barrier();
        first_stage = false;
        stamp(std::integral_constant<int, 0>{});
        // The stage's DMA goes out as one burst per wave at the top of the stage.  (Measured alternatives, DESIGN 2.2 / NOTES R6.2b: one piece
        // at a time from inside the MFMA stream -- 3-4 % slower; the two waves of a SIMD bursting in turns, the upper wave bursting for
        // both, the burst behind the first k-step -- slower or level.)  An item's FIRST stage bursts behind its first k-step: with the
        // builtin MFMAs hipcc guarded the registers the epilogue's stores read with a vmcnt(0) in front of the first MFMA that overwrote
        // them, which had to meet those stores only, not DMA pieces issued a moment ago; on the tied form the guard is gone and the burst
        // at the top of the first stage measures level (profiles/r06_kbench_w2d_first_stage_burst_at_top.txt) -- left where it is.
        if constexpr (!FIRST) issue(b_fill);
        lane_offsets();
        if constexpr (FIRST) open_kstep(b_cur);
        stamp(std::integral_constant<int, 1>{});
        if constexpr (KS == 1) {
            // one k-step per stage: it prepares the NEXT stage's (landed: the barrier above), except an item's last
            if constexpr (LAST) kstep(first_tag, std::false_type{}, b_cur, 0, b_nxt, 0);
            else kstep(first_tag, std::true_type{}, b_cur, 0, b_nxt, 0);
            if constexpr (FIRST) issue(b_fill);
            if constexpr (!FIRST && !LAST) place_ahead();
            stamp(std::integral_constant<int, 2>{});
            w2d_dma_wait<(ABL & 32768) != 0>();
            stamp(std::integral_constant<int, 4>{});
            float* t1 = b_cur; b_cur = b_nxt; b_nxt = b_fill; b_fill = t1;
            return;
        }
#ifndef AICG_EMULATED
        // ABL 262144: the upper wave of a SIMD (waves w and w + 4 share one) at issue priority 1 -- the oldest-first arbitration lets the
        // lower wave run ahead and wait ~1 900 cycles per stage at the barrier; with the priority the roles swap and the stage is as long
        if constexpr ((ABL & 262144) != 0) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
#endif
        kstep(first_tag, std::true_type{}, b_cur, 0, b_cur, 1);
        if constexpr (FIRST) issue(b_fill);
        // (steady stages only: with more than three chunks per item that is where the cursor wraps; an item's first and last stages keep
        //  their register budgets -- behind the last stage's first k-step the placement's temporaries pushed accumulators through scratch)
        if constexpr (!FIRST && !LAST) place_ahead();
        stamp(std::integral_constant<int, 2>{});
        if constexpr (LAST) kstep(std::false_type{}, std::false_type{}, b_cur, 1, b_nxt, 0);
        else kstep(std::false_type{}, std::true_type{}, b_cur, 1, b_nxt, 0);
        stamp(std::integral_constant<int, 3>{});
        w2d_dma_wait<(ABL & 32768) != 0>();
        stamp(std::integral_constant<int, 4>{});
        float* t = b_cur; b_cur = b_nxt; b_nxt = b_fill; b_fill = t;
    };
    for (int k = 0; k < my_items; ++k) {
        const int item = first + slot + k * slots;
        const int tile = item / nmu, mu = item - tile * nmu;
        // the item's first chunk starts the accumulators (MFMAs on the inline constant 0)
        if (nchunk == 1) stage(std::true_type{}, std::true_type{});
        else {
            stage(std::true_type{}, std::false_type{});
            for (int c = 1; c + 1 < nchunk; ++c) stage(std::false_type{}, std::false_type{});
            stage(std::false_type{}, std::true_type{});
        }
        lane_offsets();
        // ---- epilogue: Y = A^T M A per (row block, register); lane = (tile column l15, channel group ks) holds a 2 x 2 output block
        // per (row block, register).  Lanes l15 and l15 ^ 1 trade halves (one DPP move each way): the even lane stores the upper row
        // of both blocks -- four consecutive columns, one 16-byte store --, the odd lane the lower row: 12 stores per lane and item
        // instead of 24, each wave instruction eight 128-byte runs.
        if constexpr ((dbg & 16) != 0) {                 // (the accumulators stay live: the asm MFMAs of an unread result would be deleted)
#ifndef AICG_EMULATED
#pragma unroll
            for (int pt = 0; pt < 16; ++pt)
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) asm volatile("" :: "v"(acc[pt][rb]));
#endif
            continue;
        }
        if constexpr (TIED) w2d_mfma_settle();
        w2d_fence();   // the epilogue's index arithmetic stays out of the last k-step (scheduled into it, it pushed accumulators into scratch memory)
        const int tw_i = tile % p.tiles_w, th_i = (tile / p.tiles_w) % p.tiles_h, n = tile / (p.tiles_w * p.tiles_h);
        const int ho = th_i * NW + 2 * (wave >> 1), wo = tw_i * kW2dCols + 2 * (16 * (wave & 1) + l15);
        const long y_sc = w2d_opaque(p.y_sc), y_sh = w2d_opaque(p.y_sh);
        const float* brow = bias_s + w2d_opaque(mu) * kW2dM + 4 * ks;
        // the whole wave's 2 x 32 columns exist and rows are 16-byte aligned: the straight-line form
        const bool full = ho + 1 < p.Ho && tw_i * kW2dCols + 32 * (wave & 1) + 32 <= p.Wo && ((p.y_sn | p.y_sc | p.y_sh) & 3) == 0 &&
                          ((uintptr_t)p.y & 15) == 0;
        auto body = [&](auto act_tag, auto full_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
            constexpr bool FULL = decltype(full_tag)::value;
            const int odd = l15 & 1;
            float* ybase = p.y + (long)n * p.y_sn + (long)ho * p.y_sh + (FULL ? (long)odd * y_sh + (wo - 2 * odd) : (long)wo);
            // what one output channel stores: 2 x 2 outputs of this lane's tile
            auto store = [&](int m, float y00, float y01, float y10, float y11) __attribute__((always_inline)) {
                y00 = act_static<ACT>(y00, p.act, p.act_slope);
                y01 = act_static<ACT>(y01, p.act, p.act_slope);
                y10 = act_static<ACT>(y10, p.act, p.act_slope);
                y11 = act_static<ACT>(y11, p.act, p.act_slope);
                float* dst = ybase + (long)m * y_sc;
                if constexpr (FULL) {
                    // what the partner needs of this lane: the even lane's lower row, the odd lane's upper row
                    const float g0 = quad_xor1(odd ? y00 : y10), g1 = quad_xor1(odd ? y01 : y11);
                    const float4 v = odd ? make_float4(g0, g1, y10, y11) : make_float4(y00, y01, g0, g1);
                    if constexpr ((dbg & 128) == 0) *reinterpret_cast<float4*>(dst) = v;
                    else asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                } else {
                    const bool row1 = ho + 1 < p.Ho, col1 = wo + 1 < p.Wo;
                    if (ho < p.Ho && wo < p.Wo) {
                        dst[0] = y00;
                        if (col1) dst[1] = y01;
                        if (row1) { dst[y_sh] = y10; if (col1) dst[y_sh + 1] = y11; }
                    }
                }
            };
            if constexpr (PK_EPI) {
                // A^T M A on channel PAIRS: registers r, r + 1 of an accumulator quad are two consecutive channels of the same point, so the 28
                // additions of an output channel run as 28 v_pk_add_f32 for two channels on operands that are register pairs as they
                // stand (vectorised over the outputs of ONE channel, hipcc packed a third of the additions and paid a v_mov per packed operand:
                // 766 instructions per item, 120 of them moves).  Same additions in the same order per channel: bit-identical.
                typedef float w2d_f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    const float4 bq = *reinterpret_cast<const float4*>(brow + rb * 16);
#pragma unroll
                    for (int rp = 0; rp < 4; rp += 2) {
                        w2d_fence();   // one channel pair at a time (hoisted, the store addresses and their temporaries cost accumulator registers)
                        auto A = [&](int pt) __attribute__((always_inline)) { return w2d_f32x2{acc[pt][rb][rp], acc[pt][rb][rp + 1]}; };
                        const w2d_f32x2 b2 = rp == 0 ? w2d_f32x2{bq.x, bq.y} : w2d_f32x2{bq.z, bq.w};
                        // (a - b is v_pk_add_f32 with the negation as an operand modifier: there is no packed subtraction; rows of W = A^T M one
                        //  at a time, so that at most six pairs are live)
                        auto sub = [&](w2d_f32x2 a, w2d_f32x2 b) __attribute__((always_inline)) {
#ifdef AICG_EMULATED
                            return a - b;
#else
                            w2d_f32x2 r;          // (written a + (-b), hipcc folds it back into two scalar v_sub_f32)
                            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
                            return r;
#endif
                        };
                        auto W0 = [&](int i) __attribute__((always_inline)) { return (A(4 * i) + A(4 * i + 1)) + A(4 * i + 2); };
                        auto W1 = [&](int i) __attribute__((always_inline)) { return sub(sub(A(4 * i + 1), A(4 * i + 2)), A(4 * i + 3)); };
                        w2d_f32x2 y00 = W0(0), y01 = W1(0);
                        const w2d_f32x2 w10 = W0(1), w11 = W1(1);
                        y00 = y00 + w10; y01 = y01 + w11;
                        const w2d_f32x2 w20 = W0(2), w21 = W1(2);
                        y00 = (y00 + w20) + b2; y01 = (y01 + w21) + b2;
                        w2d_f32x2 y10 = sub(w10, w20), y11 = sub(w11, w21);
                        const w2d_f32x2 w30 = W0(3), w31 = W1(3);
                        y10 = sub(y10, w30) + b2; y11 = sub(y11, w31) + b2;
                        const int m = mu * kW2dM + rb * 16 + 4 * ks + rp;
                        store(m, y00.x, y01.x, y10.x, y11.x);
                        store(m + 1, y00.y, y01.y, y10.y, y11.y);
                    }
                }
                return;
            }
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const float4 bq = *reinterpret_cast<const float4*>(brow + rb * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    w2d_fence();   // one output channel at a time: hoisted, the twelve store addresses and their temporaries spill accumulators
                    const int m = mu * kW2dM + rb * 16 + 4 * ks + r;
                    const float bm = r == 0 ? bq.x : r == 1 ? bq.y : r == 2 ? bq.z : bq.w;
                    float Wc[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        Wc[i][0] = (acc[4 * i][rb][r] + acc[4 * i + 1][rb][r]) + acc[4 * i + 2][rb][r];
                        Wc[i][1] = (acc[4 * i + 1][rb][r] - acc[4 * i + 2][rb][r]) - acc[4 * i + 3][rb][r];
                    }
                    store(m, ((Wc[0][0] + Wc[1][0]) + Wc[2][0]) + bm, ((Wc[0][1] + Wc[1][1]) + Wc[2][1]) + bm,
                          ((Wc[1][0] - Wc[2][0]) - Wc[3][0]) + bm, ((Wc[1][1] - Wc[2][1]) - Wc[3][1]) + bm);
                }
            }
        };
        if (full) {
            if (p.act == AICG_ACT_NONE) body(std::integral_constant<int, 0>{}, std::true_type{});
            else if (p.act == AICG_ACT_RELU) body(std::integral_constant<int, 1>{}, std::true_type{});
            else body(std::integral_constant<int, 3>{}, std::true_type{});
        } else {
            if (p.act == AICG_ACT_NONE) body(std::integral_constant<int, 0>{}, std::false_type{});
            else if (p.act == AICG_ACT_RELU) body(std::integral_constant<int, 1>{}, std::false_type{});
            else body(std::integral_constant<int, 3>{}, std::false_type{});
        }
        stamp(std::integral_constant<int, 5>{});
    }
#ifndef AICG_EMULATED
    if ((dbg & (64 | 256)) != 0 && blockIdx.x == 0 && tid == 0) {
        p.y[0] = (float)(clock64() - clk0);
        p.y[1] = (float)(wall_clock64() - wall0);
    }
    if constexpr ((dbg & 256) != 0) {                // per wave of workgroup 0: y[16 + 8 wave + phase]; y[8] = its items
        if (blockIdx.x == 0 && (tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) p.y[16 + 8 * wave + k] = (float)tph[k];
            if (tid == 0) p.y[8] = (float)my_items;
        }
    }
#endif
}

// returns 0 launched, < 0 error, 1 not applicable.  p.w3 must point at the F(2 x 2, 3 x 3) image of ops.winograd2d_image:
// [Cout / 48][ceil(Cin / 8)][s = 0..1][point 0..15][ks = 0..3][m = 0..47], element = U[48 mu + m][8 chunk + 4 s + ks][point].
template <int NW, int PF, int ABL = 0, int KS = 2>
static int launch_conv_w2d(ConvArgs& p, hipStream_t stream) {
    constexpr int BUFS = 3;
    auto al4 = [](long v) { return (v & 3) == 0; };
    if (p.Cout_g % kW2dM || (p.W & 3) || !al4(p.x_sn) || !al4(p.x_sc) || !al4(p.x_sh) || ((uintptr_t)p.x & 15) || ((uintptr_t)p.w3 & 15)) return 1;
    if ((long)8 * p.x_sc + (long)p.H * p.x_sh >= (1L << 29)) return 1;   // 32-bit byte offsets inside an 8-channel slab
    p.tiles_w = idiv_up(p.Wo, kW2dCols);
    p.tiles_h = idiv_up(p.Ho, NW);
    p.nchunk = idiv_up(p.Cin_g, 8) * (KS == 1 ? 2 : 1);     // stages per item (KS 1: the second half of a ragged last chunk is zero weights)
    p.Mpad = p.Cout_g / kW2dM;
    const long nitems = (long)p.N * p.tiles_h * p.tiles_w * p.Mpad;
    if (nitems > 2147483647L - 8) return fail(AICG_E_SHAPE, "conv: too many output tiles");
    constexpr int PP = (4 * KS * w2d_plane_quads(NW) + 63) / 64;     // patch pieces per stage
    const size_t lds = (size_t)(BUFS * (kW2dWFloats * KS / 2 + PP * 256) + ((p.Cout_g + 3) & ~3) + 2 * ((PP + NW - 1) / NW) * 64 * NW) * sizeof(float);   // stages, bias, DMA decode + offsets
    if (lds > (KS == 1 ? 80 : 160) * 1024) return 1;
    const int per_xcd = (int)((nitems + 7) >> 3);
    int slots = KS == 1 ? 64 : 32;                    // workgroups per XCD: one per CU (KS 1: two)
    if (slots > per_xcd) slots = per_xcd;
    allow_dynamic_lds((const void*)conv_w2d_kernel<NW, PF, ABL, KS>, lds);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_w2d_kernel<NW, PF, ABL, KS>), dim3((unsigned)(8 * slots)), dim3(64 * NW), lds, stream, p);
    return check_launch("conv_w2d_kernel");
}

int run_w2d_8(ConvArgs& p, hipStream_t st);
int run_w2d_4(ConvArgs& p, hipStream_t st);
int run_w2d_8q(ConvArgs& p, hipStream_t st);
int run_w2d_8p(ConvArgs& p, hipStream_t st);
int run_w2d_4p2(ConvArgs& p, hipStream_t st);  // four waves, pair fragments, 4-channel stages: two workgroups per CU   // eight waves on PAIR fragments (image [s][p / 2][ks][m][p % 2])
int run_w2d_4q(ConvArgs& p, hipStream_t st);
int run_w2d_ablation(ConvArgs& p, hipStream_t st, int bits);
int run_w2d_pairs_ablation(ConvArgs& p, hipStream_t st, int bits);   // dev library only: the pair-fragment form with ABL = bits   // dev library only: the 8-wave form with ABL = bits (1 = unknown variant)

}  // namespace aicg
