// aicg_conv_forward: geometry checks, tile choice and launch of the implicit-GEMM convolution family (conv_kernels.h), plus
// the pointwise streaming kernel for 1x1 layers with <= 8 channels on one side.
#include "conv_ws3s.h"
#include "conv_ws3w.h"
#include "conv_w2d.h"
#include "conv_g1.h"
#include "conv_g1s.h"
#include "conv_g1w.h"
#ifdef AICG_DEV_SWITCHES
#include "conv_g1k.h"
#endif

namespace aicg {

#ifdef AICG_CONV_TRACE
__device__ unsigned long long g_conv_trace[kTraceWgs * kTraceSlots];
__device__ unsigned long long g_conv_trace2[kTrace2Wgs * 2 * 32 * 4];
#endif

// ---- pointwise streaming kernel -------------------------------------------------------------------------------------------
// 1x1 layers with at most 8 input or 8 output channels (MDX-Net's 4 -> 48 stem and 48 -> 4 head: 2.4 GB of activations each,
// 2 x 4 x 48 flop per position) are pure HBM streams; as MFMA tiles they ran at 0.45 TB/s.  One thread = 4 consecutive positions
// (float4); the weights are wave-uniform scalar loads from the packed [Cin_pad][Mpad] image.
template <bool FEW_IN>
__global__ void __launch_bounds__(256) conv_pointwise_kernel(ConvArgs p) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;   // float4 index within one image (H * W / 4 of them)
    const long per_img = (long)p.Ho * (p.Wo >> 2);
    if (q >= per_img) return;
    const int n = blockIdx.y;
    const int h = (int)(q / (p.Wo >> 2));
    const int w4 = (int)(q - (long)h * (p.Wo >> 2)) * 4;
    const float* xp = p.x + (long)n * p.x_sn + (long)h * p.x_sh + w4;
    float* yp = p.y + (long)n * p.y_sn + (long)h * p.y_sh + w4;
    const float* rp = p.res ? p.res + (long)n * p.r_sn + (long)h * p.r_sh + w4 : nullptr;
    auto finish = [&](float4 a, int co) __attribute__((always_inline)) {
        if (rp) {
            const float4 r = *reinterpret_cast<const float4*>(rp + (long)co * p.r_sc);
            if (p.res_first) { a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
            a.x = apply_act(a.x, p.act, p.act_slope); a.y = apply_act(a.y, p.act, p.act_slope);
            a.z = apply_act(a.z, p.act, p.act_slope); a.w = apply_act(a.w, p.act, p.act_slope);
            if (!p.res_first) { a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
        } else {
            a.x = apply_act(a.x, p.act, p.act_slope); a.y = apply_act(a.y, p.act, p.act_slope);
            a.z = apply_act(a.z, p.act, p.act_slope); a.w = apply_act(a.w, p.act, p.act_slope);
        }
        a.x *= p.out_scale; a.y *= p.out_scale; a.z *= p.out_scale; a.w *= p.out_scale;
        *reinterpret_cast<float4*>(yp + (long)co * p.y_sc) = a;
    };
    if (FEW_IN) {
        float4 xv[8];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
            xv[ci] = ci < p.Cin_g ? *reinterpret_cast<const float4*>(xp + (long)ci * p.x_sc) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int co = 0; co < p.Cout_g; ++co) {
            const float b = p.bias ? p.bias[co] : 0.f;
            float4 a = make_float4(b, b, b, b);
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                if (ci < p.Cin_g) {
                    const float wv = p.w[ci * p.Mpad + co];
                    a.x += wv * xv[ci].x; a.y += wv * xv[ci].y; a.z += wv * xv[ci].z; a.w += wv * xv[ci].w;
                }
            }
            finish(a, co);
        }
    } else {
        float4 a[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            const float b = (p.bias && co < p.Cout_g) ? p.bias[co] : 0.f;
            a[co] = make_float4(b, b, b, b);
        }
        for (int ci = 0; ci < p.Cin_g; ++ci) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + (long)ci * p.x_sc);
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                if (co < p.Cout_g) {
                    const float wv = p.w[ci * p.Mpad + co];
                    a[co].x += wv * xv.x; a[co].y += wv * xv.y; a[co].z += wv * xv.z; a[co].w += wv * xv.w;
                }
            }
        }
#pragma unroll
        for (int co = 0; co < 8; ++co)
            if (co < p.Cout_g) finish(a[co], co);
    }
}

}  // namespace aicg

using namespace aicg;

#ifdef AICG_CONV_TRACE
// tools only: copy the timeline out (host buffer of kTraceWgs * kTraceSlots u64) and clear it
extern "C" int aicg_conv_trace_read(unsigned long long* host, int clear) {
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_trace), sizeof(unsigned long long) * kTraceWgs * kTraceSlots) != hipSuccess)
        return fail(AICG_E_HIP, "trace copy failed");
    if (clear) { static unsigned long long z[kTraceWgs * kTraceSlots]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_trace), z, sizeof(z)); }
    return AICG_OK;
}
extern "C" int aicg_conv_trace2_read(unsigned long long* host) {
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_trace2), sizeof(unsigned long long) * kTrace2Wgs * 2 * 32 * 4) != hipSuccess)
        return fail(AICG_E_HIP, "trace copy failed");
    return AICG_OK;
}
#endif

extern "C" int aicg_conv_bkc(int taps) {
    // granularity to which the packed weights pad the input-channel axis (the kernel picks its K chunk <= this)
    (void)taps;
    return 32;
}

extern "C" int aicg_conv_desc_size(void) { return (int)sizeof(aicg_conv_desc); }

extern "C" int aicg_conv_forward(const aicg_conv_desc* d, const float* x, const float* w_packed, const float* bias,
                                 const float* res, float* y, void* stream) {
    if (!d || !x || !w_packed || !y) return fail(AICG_E_ARG, "aicg_conv_forward: null pointer");
    if (d->groups < 1 || d->Cin % d->groups || d->Cout % d->groups)
        return fail(AICG_E_SHAPE, "aicg_conv_forward: channels not divisible by groups");
    if (d->KH < 1 || d->KW < 1 || d->stride_h < 1 || d->stride_w < 1 || d->dil_h < 1 || d->dil_w < 1)
        return fail(AICG_E_SHAPE, "aicg_conv_forward: bad kernel geometry");
    const int pad_h_end = d->pad_h_end < 0 ? d->pad_h : d->pad_h_end, pad_w_end = d->pad_w_end < 0 ? d->pad_w : d->pad_w_end;
    int Ho = (d->H + d->pad_h + pad_h_end - d->dil_h * (d->KH - 1) - 1) / d->stride_h + 1;
    int Wo = (d->W + d->pad_w + pad_w_end - d->dil_w * (d->KW - 1) - 1) / d->stride_w + 1;
    // the caller may ask for fewer outputs than the geometry yields (fairseq SamePad drops the last frame)
    if (d->Ho > Ho || d->Wo > Wo || d->Ho < 0 || d->Wo < 0)
        return fail(AICG_E_SHAPE, "aicg_conv_forward: output %dx%d exceeds geometry (%dx%d)", d->Ho, d->Wo, Ho, Wo);
    Ho = d->Ho; Wo = d->Wo;
    if (d->N == 0 || Ho <= 0 || Wo <= 0) return AICG_OK;
    ConvArgs p;
    p.stagger = p.stagger_first = p.wide_ok = 0;
    p.x = x; p.w = w_packed; p.bias = bias; p.res = res; p.y = y;
    p.N = d->N; p.Cin_g = d->Cin / d->groups; p.H = d->H; p.W = d->W; p.Cout_g = d->Cout / d->groups;
    p.Ho = Ho; p.Wo = Wo; p.KH = d->KH; p.KW = d->KW; p.sh = d->stride_h; p.sw = d->stride_w;
    p.ph = d->pad_h; p.pw = d->pad_w; p.dh = d->dil_h; p.dw = d->dil_w; p.groups = d->groups;
    p.x_sn = d->x_sn; p.x_sc = d->x_sc; p.x_sh = d->x_sh;
    p.y_sn = d->y_sn; p.y_sc = d->y_sc; p.y_sh = d->y_sh;
    p.r_sn = d->r_sn; p.r_sc = d->r_sc; p.r_sh = d->r_sh;
    p.pre_act = d->pre_act; p.pre_slope = d->pre_slope; p.act = d->act; p.act_slope = d->act_slope;
    p.out_scale = d->out_scale; p.accumulate = d->accumulate; p.res_first = d->res_before_act;
    p.shuffle = d->shuffle; p.res_mul = d->res_mul;
    if (p.shuffle != 0 && (p.shuffle != 2 || d->Cout % 4 || d->groups != 1))
        return fail(AICG_E_ARG, "aicg_conv_forward: shuffle must be 0 or 2 (Cout %% 4 == 0, groups == 1)");
    if (p.res_mul && (!res || d->res_before_act)) return fail(AICG_E_ARG, "aicg_conv_forward: res_mul needs res and res_before_act == 0");
    p.taps = p.KH * p.KW;
    p.Mpad = idiv_up(p.Cout_g, 32) * 32;
    p.Cin_pad = idiv_up(p.Cin_g, 32) * 32;
    p.w_group_stride = (long)p.taps * p.Cin_pad * p.Mpad;
    p.w3 = d->packed_v3 ? w_packed + (long)p.groups * p.w_group_stride : nullptr;
    p.wsplit = d->packed_v3 && d->split == 1 ? w_packed + 2L * p.groups * p.w_group_stride : nullptr;
    p.f16 = d->split == 2;

    if (d->wino == 8) {
        // one-dimensional Winograd F(2, 3) of a k = 3 / 7 / 11, dilation-1 layer (conv_g1w.h): w_packed is the image pair of the
        // (Cout, Cin, 1, S) slot kernel, S = 4 / 10 / 15 (ops.winograd1d_kernel)
        if (!d->packed_v3 || (p.KW != 3 && p.KW != 5 && p.KW != 7 && p.KW != 11))
            return fail(AICG_E_ARG, "aicg_conv_forward: wino 8 needs a packed k = 3 / 5 / 7 / 11 one-dimensional layer");
        const int nslot = p.KW == 3 ? 4 : p.KW == 5 ? 7 : p.KW == 7 ? 10 : 15;
        p.w3 = w_packed + (long)nslot * p.Cin_pad * p.Mpad;
        if (!conv_g1w_applicable(p, pad_w_end))
            return fail(AICG_E_ARG, "aicg_conv_forward: wino 8 needs stride 1, dilation 1 / 3 / 5, same padding, one group, W %% 4 == 0, 16-byte aligned "
                                    "rows (strides %% 4 == 0), no input activation but a leaky ReLU");
        hipStream_t st8 = (hipStream_t)stream;
        // Tile 32 x 512 -- all four waves share the tile's 32 rows, 512 outputs amortise a unit's weights: it won on every layer of the
        // vocoder, 32 to 256 channels (profiles/r05_kbench_g1w_*.txt).  Development builds carry the measured-and-lost variants for A/B:
        // aicg_conv_desc.gemm_tile 2 = the 64 x 256 tile, 5 = explicit MFMA / VALU interleave, 6 = persistent tile walk (2-5 % slower:
        // profiles/r05_kbench_g1w_persistent.txt), 7 = that walk on three workgroups (tests) -- all for dilation 1
        int rc;
        p.dbg = 0;
#ifdef AICG_DEV_SWITCHES
        p.dbg = d->gemm_tile == 7 ? 3 : 0;
        if (p.dw == 1 && d->gemm_tile == 2) rc = run_g1w_64x256(p, st8);
        else if (p.dw == 1 && d->gemm_tile == 5) rc = run_g1w_32x512_sched(p, st8);
        else if (p.dw == 1 && (d->gemm_tile == 6 || d->gemm_tile == 7)) rc = run_g1w_32x512_pers(p, st8);
        else
#endif
            rc = p.f16 ? run_g1w_32x512_h(p, st8) : run_g1w_32x512(p, st8);
        if (rc == 1) return fail(AICG_E_SHAPE, "aicg_conv_forward: wino 8 layer does not fit the kernel's LDS budget");
        return rc;
    }
    if (d->wino) {
        // Winograd F(2, 3) along rows (conv_ws3w.h): w_packed is the image pair of the (Cout, Cin, 3, 4) kernel, 12 taps;
        // wino == 2: F(2 x 2, 3 x 3) (conv_w2d.h): w_packed is the [Cout / 48][Cin / 8][2][16][4][48] image of U = G g G^T
        if (p.KH != 3 || p.KW != 3 || p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1 || p.ph != 1 || p.pw != 1 || pad_h_end != 1 ||
            pad_w_end != 1 || p.groups != 1 || res || p.accumulate || p.shuffle || p.res_mul || p.pre_act != AICG_ACT_NONE ||
            p.out_scale != 1.f || (!d->packed_v3 && d->wino < 2) || (p.act != AICG_ACT_NONE && p.act != AICG_ACT_RELU) || Ho != p.H || Wo != p.W)
            return fail(AICG_E_ARG, "aicg_conv_forward: wino needs a plain 3x3 / stride 1 / padding 1 layer (bias + none|ReLU epilogue)");
        if (d->wino >= 2) {
            AICG_SWITCH(w2d_ablate, "AICG_CONV_ABLATE", 0);
            p.dbg = w2d_ablate;
            p.w3 = w_packed;
            // 2: eight waves (two per SIMD) on an 8 x 64 tile; 3: four waves (one per SIMD) on a 4 x 64 tile; 4 / 5: the same with the
            // quad-fragment image ([s][p / 4][ks][m][p % 4]: one 16-byte fragment read per four MFMAs)
            hipStream_t st2 = (hipStream_t)stream;
            if (w2d_ablate && (d->wino == 2 || d->wino == 12)) {
                const int ra = d->wino == 2 ? run_w2d_ablation(p, st2, (int)w2d_ablate) : run_w2d_pairs_ablation(p, st2, (int)w2d_ablate);
                if (ra != 1) return ra;
                return fail(AICG_E_ARG, "aicg_conv_forward: AICG_CONV_ABLATE=%ld is not an instantiated variant of conv_w2d (or not the dev library)", (long)w2d_ablate);
            }
#ifdef AICG_DEV_SWITCHES
            // dev library, for round-robin A/B in one process (tools/kbench_w2d_ab.py): 6 = the eight-wave form with the round-5 stage burst
            if (d->wino == 6) return run_w2d_ablation(p, st2, 512);
            if (d->wino == 7) return run_w2d_ablation(p, st2, 1024);   // patch pieces from one L2-resident KiB (wrong results)
            if (d->wino == 9) return run_w2d_ablation(p, st2, 2048);   // no patch pieces at all (wrong results)
            if (d->wino == 10) return run_w2d_ablation(p, st2, 1);     // no DMA at all (wrong results)
            if (d->wino == 14) return run_w2d_pairs_ablation(p, st2, 16384);  // pair fragments, MFMAs through the builtin (untied destinations: accumulator quads through scratch memory)
            if (d->wino == 15) return run_w2d_pairs_ablation(p, st2, 65536);  // pair fragments, the epilogue one output channel at a time
            if (d->wino == 13) return run_w2d_pairs_ablation(p, st2, 8192);   // pair fragments, place() at the top of the stage that needs it
#endif
            // 16 (development builds): four waves on pair fragments with 4-channel stages -- two workgroups per CU;
            // 12: eight waves on the PAIR-fragment image ([s][p / 2][ks][m][p % 2]: one 8-byte fragment read per two MFMAs)
            const int rc = d->wino == 2 ? run_w2d_8(p, st2) : d->wino == 3 ? run_w2d_4(p, st2) : d->wino == 4 ? run_w2d_8q(p, st2)
                         : d->wino == 12 ? run_w2d_8p(p, st2) : d->wino == 16 ? run_w2d_4p2(p, st2) : run_w2d_4q(p, st2);
            if (rc == 1)
                return fail(AICG_E_ARG, "aicg_conv_forward: wino 2 needs Cout %% 48 == 0, W %% 4 == 0, 16-byte aligned x with strides %% 4 == 0");
            return rc;
        }
        AICG_SWITCH(wino_ablate, "AICG_CONV_ABLATE", 0);
        p.dbg = wino_ablate;
        p.taps = 12;
        p.w_group_stride = (long)p.taps * p.Cin_pad * p.Mpad;
        p.w3 = w_packed + p.w_group_stride;
        // output-channel tile with the least padding (the larger on ties): 96 / 64 / 32 rows on the 32 x 32 x 2 MFMA, 48 on 16 x 16 x 4
        const int M = p.Cout_g;
        int bm = 32;
        long best = 1L << 40;
        const int cands[4] = {96, 64, 48, 32};
        AICG_SWITCH(wino_bm, "AICG_WINO_BM", 0);
        for (int i = 0; i < 4; ++i) {
            const long padded = (long)idiv_up(M, cands[i]) * cands[i];
            if (padded < best) { best = padded; bm = cands[i]; }
        }
        if (wino_bm) bm = (int)wino_bm;
        hipStream_t wst = (hipStream_t)stream;
        auto run = [&](ConvArgs& a, int tile) {
            return tile == 96 ? run_ws3w_96(a, wst) : tile == 64 ? run_ws3w_64(a, wst) : tile == 48 ? run_ws3w_48(a, wst) : run_ws3w_32(a, wst);
        };
        int rc;
        AICG_SWITCH(wino_split, "AICG_WINO_SPLIT", 1);
        const long tiles4 = (long)p.N * idiv_up(Ho, 4) * idiv_up(Wo, 64);
        AICG_SWITCH(wino_split_tiles, "AICG_WINO_SPLIT_TILES", 1024);   // tests lower it
        if (wino_split && !wino_bm && M > 48 && M % 96 == 48 && tiles4 >= wino_split_tiles) {
            // 144 / 240 / ... rows on a map large enough to fill the chip twice over: 96-row tiles (107 executed TFLOP/s) for all but the
            // last 48 rows, which take the 48-row kernel (95) -- MDX level 2: 2.04 -> 1.82 ms; on the small level 4 two launches lose.
            // The second launch sees the same layer through shifted pointers: row m of it is row M - 48 + m of the images
            ConvArgs a = p, b = p;
            a.Cout_g = M - 48;
            b.Cout_g = 48;
            if (b.bias) b.bias += M - 48;
            b.y += (long)(M - 48) * p.y_sc;
            b.w3 += (long)(M - 48) * 4;   // [tap][chunk][parity][Mpad][4]: rows are the float4 index
            rc = run(a, 96);
            if (rc == 0) rc = run(b, 48);
        } else {
            rc = run(p, bm);
        }
        if (rc == 1) return fail(AICG_E_SHAPE, "aicg_conv_forward: wino layer too large for the kernel's 32-bit offsets");
        return rc;
    }

    // pointwise streaming form: 1x1, unit stride, no padding, <= 8 channels on one side, float4-aligned rows
    {
        AICG_SWITCH(pointwise, "AICG_CONV_POINTWISE", 1);
        const bool few_in = p.Cin_g <= 8, few_out = p.Cout_g <= 8;
        auto al4 = [](long v) { return (v & 3) == 0; };
        const bool aligned = (Wo & 3) == 0 && al4(p.x_sn) && al4(p.x_sc) && al4(p.x_sh) && al4(p.y_sn) && al4(p.y_sc) && al4(p.y_sh) &&
                             (!res || (al4(p.r_sn) && al4(p.r_sc) && al4(p.r_sh) && ((uintptr_t)res & 15) == 0)) &&
                             ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0;
        if (pointwise && p.taps == 1 && p.groups == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && pad_h_end == 0 &&
            pad_w_end == 0 && (few_in || few_out) && p.Cin_g <= 512 && p.Cout_g <= 512 && p.pre_act == AICG_ACT_NONE &&
            !p.accumulate && aligned && Ho == p.H && Wo == p.W && !p.shuffle && !p.res_mul) {
            const long per_img = (long)Ho * (Wo >> 2);
            dim3 grid((unsigned)ldiv_up(per_img, 256), (unsigned)p.N);
            if (few_in) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_pointwise_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, p);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_pointwise_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, p);
            return check_launch("conv_pointwise_kernel");
        }
    }

    // LDS-DMA staged GEMM (conv_g1.h): 1 x 1 layers over contiguous, 16-byte-aligned maps, no input activation but a leaky ReLU, fp32 path.
    // aicg_conv_desc.gemm_tile (or, dev builds, AICG_CONV_G1): 0 the policy below, 1 off, 2 / 3 / 4 force the 128 x 256 / 64 x 256 /
    // 192 x 256 tile (dev: 5 / 6 the 256 x 256 / 128 x 512 probes)
    {
        AICG_SWITCH(g1env, "AICG_CONV_G1", 0);
        const long g1 = d->gemm_tile ? d->gemm_tile : g1env;
        if (g1 != 1 && !p.wsplit && p.Cout_g > 32 && conv_g1_applicable(p, pad_h_end, pad_w_end)) {
            const long HW = (long)p.H * p.W;
            const int M = p.Cout_g;
            auto wgs = [&](int bm) { return (long)p.N * idiv_up(M, bm) * ldiv_up(HW, 256); };
            hipStream_t gst = (hipStream_t)stream;
            int rc = 1;
            if (p.f16 && !p.shuffle) {
                // fp16 operands: a quarter of the MFMA time per tile -- the 64-row tile (three workgroups per CU) unless 128 rows divide
                // the layer better
                const long t64 = wgs(64), t128 = wgs(128);
                if (g1 == 3 || (g1 != 2 && ldiv_up(t64, 768) * 64 < ldiv_up(t128, 512) * 128)) rc = run_g1_64x256_h(p, gst);
                else rc = run_g1_128x256_h(p, gst);
            }
            else if (g1 == 2) rc = run_g1_128x256(p, gst);
            else if (g1 == 3) rc = run_g1_64x256(p, gst);
            else if (g1 == 4) rc = run_g1_192x256(p, gst);
#ifdef AICG_DEV_SWITCHES
            else if (g1 == 5) rc = run_g1_256x256(p, gst);
            else if (g1 == 6) rc = run_g1_128x512(p, gst);
            else if (g1 >= 12 && g1 <= 14) rc = run_g1_burst(p, gst, (int)g1 - 10);
#endif
            else {
                // Measured (tools/kbench_g1.py, profiles/r04_kbench_g1.txt: 24 shapes x 3 tiles, round-robin): every tile runs at ~0.8 of
                // the MFMA time of the rows it covers, and a launch takes as long as its busiest CU -- ceil(tiles / 256) tiles of BM rows,
                // whether they share the CU's SIMDs (two / three workgroups per CU) or follow each other.  So: the tile with the least
                // ceil(tiles / 256) x BM, padded rows included; the 64-row tile pays ~4 % for its DMA traffic, the 192-row one gains ~2 %
                // but runs one workgroup per CU with its epilogue exposed -- not for short K (the output stream dominates there).
                // Fewer than 256 workgroups of 64 rows: the small tiles of conv_ws3 fill the chip better.
                const long t64 = wgs(64);
                if (t64 >= 256) {
                    auto cost = [&](int bm, double f) { return (double)ldiv_up(wgs(bm), 256) * bm * f; };
                    const double c64 = cost(64, 1.04), c128 = cost(128, 1.0);
                    const double c192 = (M % 192 == 0 && p.Cin_g >= 256) ? cost(192, 0.98) : 1e30;
                    if (c192 <= c64 && c192 <= c128) rc = run_g1_192x256(p, gst);
                    else if (c128 <= c64) rc = run_g1_128x256(p, gst);
                    else rc = run_g1_64x256(p, gst);
                }
            }
            if (rc <= 0) return rc;
        }
    }

    // LDS-DMA staged STRIDE-2 k-tap 1-D convolution (conv_g1s.h: k = 2 / 3, no padding, 16-byte aligned rows on both sides): HuBERT's feature
    // extractor.  aicg_conv_desc.gemm_tile: 0 the policy, 1 off, 2 / 3 force the 128 x 256 / 64 x 256 tile.  Policy, measured round-robin
    // on the extractor's six layers (tools/kbench_g1s.py, profiles/r05_kbench_g1s.txt): the 64-row tile (three workgroups per CU) wins on
    // every layer -- 138.6 / 128.1 / 108.2 / 106.2 TFLOP/s on the k = 3 layers against 128.6 / 111.6 / 107.9 / 103.5 for the 128-row tile and
    // 111.3 / 103.0 / 92.5 / 78.2 for conv_ws3 -- down to ~200 workgroups (k = 2 on 13 201 frames: 83 us against 99); the last layer
    // (104 workgroups) stays on conv_ws3's smaller tiles (55 us against 83).
    if (d->gemm_tile != 1 && !p.wsplit && p.Cout_g > 32 && conv_g1s_applicable(p, pad_w_end)) {
        const long w64 = (long)p.N * idiv_up(p.Cout_g, 64) * idiv_up(p.Wo, 256);
        int rc = 1;
        if (d->gemm_tile == 2) rc = run_g1s_128x256(p, (hipStream_t)stream);
        else if (d->gemm_tile == 3 || w64 >= 160) rc = run_g1s_64x256(p, (hipStream_t)stream);
        if (rc <= 0) return rc;
    }

#ifdef AICG_DEV_SWITCHES
    // LDS-DMA staged k-tap 1-D convolution (conv_g1k.h): DEVELOPMENT BUILDS ONLY (tools' private library, the CPU emulator) -- measured on
    // the vocoder's ResBlock layers (profiles/r04_kbench_g1k.txt) conv_ws3 wins by 10-25 %, so the product library neither routes to it
    // nor carries it.  aicg_conv_desc.gemm_tile 12 / 13 force its 128 x 256 / 64 x 256 tile (codes of their own: 2 / 3 / 4 select conv_g1
    // tiles on 1 x 1 layers and must not reroute the k-tap layers).
    if (d->gemm_tile == 12 || d->gemm_tile == 13) {
        if (!p.wsplit && p.Cout_g > 32 && conv_g1k_applicable(p, pad_w_end)) {
            const int rc = d->gemm_tile == 12 ? run_g1k_128x256(p, (hipStream_t)stream) : run_g1k_64x256(p, (hipStream_t)stream);
            if (rc <= 0) return rc;
        }
    }
#endif

    // tile shape: the BM in {160, 128, 96, 64, 32} with the least padded M (larger BM on ties: one input patch
    // staging feeds more MFMAs), then enough workgroups to fill 256 CUs
    const int M = p.Cout_g;
    int BM = 32;
    {
        long best = 1L << 40;
        const int cands[5] = {160, 128, 96, 64, 32};
        for (int i = 0; i < 5; ++i) {
            const long padded = (long)idiv_up(M, cands[i]) * cands[i];
            if (padded < best) { best = padded; BM = cands[i]; }
        }
        AICG_SWITCH(force_bm, "AICG_CONV_FORCE_BM", 0);
        if (force_bm) BM = (int)force_bm;
    }
    const long npos = (long)p.N * Ho * Wo;
    hipStream_t st = (hipStream_t)stream;
    AICG_SWITCH(ablate, "AICG_CONV_ABLATE", 0);
    p.dbg = ablate;
    auto blocks = [&](int bm, int bn) { return (long)idiv_up(M, bm) * p.groups * ldiv_up(npos, bn); };
    // (AICG_CONV_WANT: tests lower the fill target so that small problems exercise the large tiles on the CPU emulator)
    AICG_SWITCH(want, "AICG_CONV_WANT", 512);
    // opt-in split precision (aicg_conv_desc.split, conv_ws3s.h): no 160-row tile there (5 x 16 accumulators leave no room for
    // hi + lo fragments), those layers take the least-padded of the other tiles
    if (p.wsplit && p.Cin_g >= 16 && M > 16) {
        int bm = 32;
        {
            long best = 1L << 40;
            const int cands[4] = {128, 96, 64, 32};
            for (int i = 0; i < 4; ++i) {
                const long padded = (long)idiv_up(M, cands[i]) * cands[i];
                if (padded < best) { best = padded; bm = cands[i]; }
            }
        }
        int rc = 1;
        // (measured: 64 x 256 with 2 x 2 tiles per consumer wave -- 12 MFMAs per 8 fragment reads -- beats 64 x 128 by 10-25 % on the
        //  64-channel vocoder stage; the same wave shape on the 128-row tile loses 3 % to the 4 x 1 one)
        if (bm == 128 && blocks(128, 128) >= want) rc = run_ws3s_128x128(p, st);
        else if (bm == 96 && blocks(96, 128) >= want) rc = run_ws3s_96x128(p, st);
        else if (M > 32) {
            if (blocks(64, 256) >= want) rc = run_ws3s_64x256(p, st);
            if (rc == 1 && blocks(64, 128) >= want) rc = run_ws3s_64x128(p, st);
        }
        if (rc == 1) {
            if (M > 32) {
                if (!(blocks(64, 128) >= want) && (blocks(64, 64) >= want || M > 64)) rc = run_ws3s_64x64(p, st);
            } else {
                if (blocks(32, 256) >= want) rc = run_ws3s_32x256(p, st);
                if (rc == 1) rc = run_ws3s_32x128(p, st);   // (also: a 256-position patch too large to stage)
            }
        }
        if (rc <= 0) return rc;
    }
    // narrow layers (16 / 48 output channels, at least 3 input channels, enough positions): 16x16x4 MFMA tiles
    AICG_SWITCH(use16, "AICG_CONV_M16", 1);
    // (measured r2: MDX level 0 97 vs 105 TFLOP/s, RMVPE level 0 57 vs 60 against conv_ws16_kernel -- 12 MFMAs per k-step already
    //  amortise the fragment hand-over there, and the 16-channel groups need the larger patch: off by default)
    AICG_SWITCH(v3m16, "AICG_CONV_V3M16", 0);
#ifdef AICG_DEV_SWITCHES
    if (use16 && v3m16 && p.w3 && p.Cin_g >= 16 && npos >= 256L * 256) {
        int rc = 1;
        if (M > 32 && M <= 48) rc = run_ws3m16_48(p, (hipStream_t)stream);
        else if (M <= 16) rc = run_ws3m16_16(p, (hipStream_t)stream);
        if (rc <= 0) return rc;
    }
#else
    (void)v3m16;
#endif
    // 8-byte fragments (conv_ws3m16h_kernel): the same 8-row K groups as the 32 x 32 kernels, half the hand-overs of conv_ws16_kernel
    AICG_SWITCH(m16h, "AICG_CONV_M16H", 1);
    if (use16 && m16h && p.w3 && p.Cin_g >= 8 && npos >= 256L * 256) {
        int rc = 1;
        if (M > 32 && M <= 48) rc = run_ws3m16h_48(p, st);
        else if (M <= 16) rc = run_ws3m16h_16(p, st);
        if (rc <= 0) return rc;
    }
    if (use16 && p.Cin_g >= 3 && npos >= 256L * 256) {
        int rc = 1;
        if (M > 32 && M <= 48) rc = run_m16_48(p, st);
        else if (M <= 16) rc = run_m16_16(p, st);
        if (rc <= 0) return rc;   // launched (0) or failed with an error code (< 0); 1 = not applicable
    }
    // A launch should give each of the 256 CUs at least ~2 workgroups: shrink the tile for small problems
    // (HuBERT / enc_p GEMMs over a few thousand frames), M first (keeps the wide, coalesced N tile), then N.
    AICG_SWITCH(ws, "AICG_CONV_WS", 1);
    // 16-byte-fragment kernels (conv_ws3.h): layers with >= 8 input channels per group
    AICG_SWITCH(v3, "AICG_CONV_V3", 1);
    if (ws && v3 && p.w3 && p.Cin_g >= 8) {
        int rc = 1;
        AICG_SWITCH(v3_160, "AICG_CONV_V3_160", 1);
        // (the shuffle / multiplicative-skip instantiation of the 160-row tile spills inside its K loop: those layers stay classic)
        if (BM == 160 && v3_160 && !p.shuffle && !p.res_mul && blocks(160, 128) >= want) rc = run_ws3_160x128(p, st);
        else if (BM == 128 && blocks(128, 128) >= want) rc = run_ws3_128x128(p, st);
        else if (BM == 96 && blocks(96, 128) >= want) rc = run_ws3_96x128(p, st);
        else if (BM != 160 && M > 32 && blocks(64, 128) >= want) rc = run_ws3_64x128(p, st);
        if (rc == 1 && BM != 160) {
            if (M > 32) {
                if (!(blocks(64, 128) >= want) && (blocks(64, 64) >= want || M > 64)) rc = run_ws3_64x64(p, st);
            } else if (blocks(32, 256) >= want) rc = run_ws3_32x256(p, st);
            else rc = run_ws3_32x128(p, st);
        }
        if (rc <= 0) return rc;
    }
    if (ws) {
        int rc = 1;
        // v2 shapes: four MFMAs per consumer k-step (r2 timeline: the K loop of the 2-MFMA shapes keeps the matrix pipe 66 % busy)
        AICG_SWITCH(v2, "AICG_CONV_V2", 0);
#ifdef AICG_DEV_SWITCHES
        if (v2 & 1) { if (BM == 128 && blocks(128, 128) >= want) rc = run_ws_128x128_k32(p, st); }
        if (rc == 1 && (v2 & 2)) { if (M > 32 && M <= 64 && blocks(64, 256) >= want) rc = run_ws_64x256(p, st); }
        if (rc == 1 && (v2 & 4)) { if (M <= 32 && blocks(32, 512) >= want) rc = run_ws_32x512(p, st); }
        if (rc <= 0) return rc;
#else
        (void)v2;
#endif
        if (BM == 160 && blocks(160, 128) >= want) rc = run_ws_160x128(p, st);
        else if (BM == 128 && blocks(128, 128) >= want) {
            // >= 2 M tiles of 128: four-consumer 64-row tiles (two workgroups per CU overlap their prologue / epilogue) measured
            // 7 % faster than the one-per-CU 128 x 128 shape on the 256-channel vocoder stage; a single 128-row tile keeps the latter
            AICG_SWITCH(bm128e, "AICG_CONV_BM128", 0);
            const bool use64 = bm128e ? bm128e == 64 : M >= 256;
            rc = use64 ? run_ws_64x128(p, st) : run_ws_128x128(p, st);
        }
        else if (BM == 96 && blocks(96, 128) >= want) rc = run_ws_96x128(p, st);
        else if (M > 32 && blocks(64, 128) >= want) rc = run_ws_64x128(p, st);
        if (rc <= 0) return rc;
        // small problems (few output positions: HuBERT / enc_p / flow GEMMs) and 32-channel layers: the same kernel on smaller tiles
        AICG_SWITCH(smallws, "AICG_CONV_SMALLWS", 1);
        if (smallws) {
            if (M > 32) {
                if (!(blocks(64, 128) >= want) && (blocks(64, 64) >= want || M > 64)) rc = run_ws_64x64(p, st);
            } else if (blocks(32, 256) >= want) rc = run_ws_32x256(p, st);
            else rc = run_ws_32x128(p, st);
            if (rc <= 0) return rc;
        }
    }
    // single-role kernels: layers the wave-specialised form could not stage.  A patch that is too large even here (wide
    // strided / dilated 2-D windows) is retried on the narrowest tile (64 positions) before giving up.
    AICG_SWITCH(eight, "AICG_CONV_8WAVE", 1);
    auto single_role = [&]() -> int {
        if (BM == 160 && blocks(160, 128) >= want) return run_sr_160x128(p, st);
#ifdef AICG_DEV_SWITCHES
        if (BM == 128 && blocks(128, 128) >= want && !eight) return run_sr_128x128_4w(p, st);
#else
        (void)eight;
#endif
        if (BM == 128 && blocks(128, 128) >= want) return run_sr_128x128_8w(p, st);
        if (BM == 96 && blocks(96, 128) >= want) return run_sr_96x128(p, st);
        if (M > 32) {
            if (blocks(64, 128) >= want) return run_sr_64x128(p, st);
            if (blocks(64, 64) >= want || M > 64) return run_sr_64x64(p, st);
        }
        if (blocks(32, 256) >= want) return run_sr_32x256(p, st);
        return run_sr_32x128(p, st);
    };
    int rc = single_role();
    if (rc == AICG_E_LDS) rc = run_sr_64x64(p, st);
    return rc;
}
