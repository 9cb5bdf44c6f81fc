/*
 * aicg.h -- C ABI of libaicg_hip.so: the MI355X (gfx950) kernels behind AICoverGen's
 * voice-conversion / MDX-Net hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors on the Python side);
 *     the library never allocates user-visible memory; scratch is passed in by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it,
 *     nothing synchronises;
 *   - return value: AICG_OK (0) or a negative AICG_E_* code; aicg_last_error() gives the message of
 *     the last failure on the calling thread;
 *   - tensors are fp32, "channel-major": (N, C, H, W) with W contiguous; 1-D signals are (N, C, 1, T);
 *   - file:line citations are relative to the reference checkout (SociallyIneptWeeb/AICoverGen).
 */
#ifndef AICG_H_
#define AICG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AICG_OK 0
#define AICG_E_SHAPE (-1) /* unsupported / inconsistent shape */
#define AICG_E_ARG (-2)   /* null pointer, bad enum */
#define AICG_E_HIP (-3)   /* a HIP launch failed */
#define AICG_E_LDS (-4)   /* tile does not fit the 160 KiB LDS */

/* activation codes used by conv epilogues / elementwise kernels */
#define AICG_ACT_NONE 0
#define AICG_ACT_RELU 1
#define AICG_ACT_LRELU 2 /* slope parameter */
#define AICG_ACT_GELU 3  /* exact erf form (fairseq / HF HuBERT "gelu") */
#define AICG_ACT_TANH 4
#define AICG_ACT_SIGMOID 5
#define AICG_ACT_LOGCLAMP 6 /* log(max(v, slope)): log-mel of rmvpe.MelSpectrogram (src/rmvpe.py:324) */

const char* aicg_last_error(void);
/* Diagnostic: name of the kernel family the calling thread's most recent entry point launched ("" before the first launch) */
const char* aicg_last_launch(void);
/* ABI version; bumped whenever a signature changes */
int aicg_abi_version(void);
/* Diagnostic: pure v_mfma_f32_32x32x2_f32 issue loop (n_blocks x 256 threads, 4*iters MFMAs per wave) to calibrate the
 * attainable fp32-MFMA rate of the device under load; out: n_blocks*256 floats. */
int aicg_mfma_probe(float* out, int n_blocks, int iters, float seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Framed STFT / iSTFT  (replaces torch.stft / torch.istft as used by MDXModel.stft / .istft,
 * src/mdx.py:37-54, and by rmvpe.MelSpectrogram.forward, src/rmvpe.py:305-314)
 *
 * Real FFT of n_fft points (n_fft = 2^a 3^b 5^c, even, <= 16384) per frame, evaluated as a complex
 * Stockham FFT of n_fft/2 points in LDS plus a split/merge step.  Tables are supplied by the caller:
 *   window  [n_fft]            analysis / synthesis window (periodic Hann for MDX)
 *   tw_half [n_fft/2][2]       exp(-2 pi i k / (n_fft/2))
 *   tw_full [n_fft/2 + 1][2]   exp(-2 pi i k / n_fft)
 * ---------------------------------------------------------------------------------------------- */

/* x: [n_sig][L] real signals.  Frame t covers x[t*hop - n_fft/2 .. +n_fft) with reflect padding
 * (torch.stft center=True).  Bin k < n_bins_out of frame t of signal s is written at
 *   out[s*o_sig + k*o_bin + t*o_frame]          (real part)
 *   out[s*o_sig + k*o_bin + t*o_frame + o_im]   (imaginary part)
 * MDXModel.stft layout (B,4,dim_f,dim_t): o_sig = 2*dim_f*dim_t, o_im = dim_f*dim_t, o_bin = dim_t, o_frame = 1. */
int aicg_stft(const float* x, float* out, const float* window, const float* tw_half, const float* tw_full,
              int n_sig, int L, int n_fft, int hop, int n_frames, int n_bins_out,
              int64_t o_sig, int64_t o_im, int64_t o_bin, int64_t o_frame, void* stream);

/* Inverse: spec addressed like aicg_stft's out (bins >= n_bins_in are zero: MDXModel.freq_pad,
 * src/mdx.py:35,46-47; imaginary parts of bin 0 and bin n_fft/2 are ignored like a C2R transform).
 * Step 1 writes windowed time frames [n_sig][n_frames][n_fft] into `frames` (caller scratch);
 * step 2 overlap-adds them, divides by the overlap-added window^2 and crops n_fft/2 each side
 * (torch.istft center=True), out: [n_sig][L]. */
int aicg_istft_frames(const float* spec, float* frames, const float* window, const float* tw_half,
                      const float* tw_full, int n_sig, int n_fft, int n_frames, int n_bins_in,
                      int64_t i_sig, int64_t i_im, int64_t i_bin, int64_t i_frame, void* stream);
int aicg_istft_ola(const float* frames, const float* window, float* out, int n_sig, int L, int n_fft,
                   int hop, int n_frames, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on the matrix cores (fp32 MFMA, exact fp32 accumulate).
 * One entry point for Conv1d / Conv2d / grouped / strided / dilated / 1x1 ("linear") layers:
 *   torch.nn.Conv1d / Conv2d / Linear as used by infer_pack (src/infer_pack/models.py:445-516,
 *   modules.py:188-213,299-312, attentions.py:190-193,387-388), rmvpe.ConvBlockRes (src/rmvpe.py:23-58),
 *   fairseq HuBERT (call site src/vc_infer_pipeline.py:398-406) and the MDX-Net graph (src/mdx.py:74-77).
 *
 *   y[n,co,ho,wo] = [y +] out_scale * ( act( bias[co] + sum_{ci,kh,kw} W[co,ci,kh,kw] *
 *                        pre_act(x[n, ci, ho*stride_h - pad_h + kh*dil_h, wo*stride_w - pad_w + kw*dil_w]) )
 *                                       + res[n,co,ho,wo] )
 * 1-D convolutions use H = KH = 1.  Strides are in elements; W is contiguous for x, y and res.
 * Weights must be packed with the layout of aicovergen_amd.ops.pack_conv_weight (per group:
 * [KH*KW][Cin_pad][Mpad], Cin_pad = roundup(Cin_g, 32), Mpad = roundup(Cout_g, 32)).
 * ---------------------------------------------------------------------------------------------- */
typedef struct aicg_conv_desc {
    int32_t N, Cin, H, W;
    int32_t Cout, Ho, Wo;
    int32_t KH, KW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups;
    int64_t x_sn, x_sc, x_sh;
    int64_t y_sn, y_sc, y_sh;
    int64_t r_sn, r_sc, r_sh;
    int32_t pre_act;   /* AICG_ACT_* applied to x while it is staged (e.g. leaky-ReLU of ResBlock1) */
    float pre_slope;
    int32_t act;       /* AICG_ACT_* applied to acc + bias */
    float act_slope;
    float out_scale;
    int32_t accumulate; /* nonzero: add into y instead of overwriting */
    int32_t res_before_act; /* nonzero: y = out_scale * act(acc + bias + res)  (e.g. emb_phone + emb_pitch -> lrelu) */
    int32_t pad_h_end, pad_w_end; /* zero padding after the last row / column; -1 = same as pad_h / pad_w
                                     (torchcrepe pads (31, 32) around its k=64 convs) */
    int32_t shuffle;              /* 0, or 2: the layer is the 1x1 GEMM of a kernel = stride = 2 ConvTranspose2d
                                     (mdx U-Net `us.*`): GEMM row m at position (ho, wo) is stored to
                                     y[n][m >> 2][2 ho + ((m >> 1) & 1)][2 wo + (m & 1)]; y (and res) strides describe that
                                     (N, Cout/4, 2 Ho, 2 Wo) tensor, bias has Cout entries */
    int32_t res_mul;              /* nonzero: y = out_scale * (act(acc + bias) * res) -- multiplicative U-Net skip */
    int32_t packed_v3;            /* nonzero: w_packed holds two images of the weights back to back, each
                                     groups * KH*KW * Cin_pad * Mpad floats: the classic [tap][Cin_pad][Mpad] one and the
                                     k8-interleaved [tap][Cin_pad/8][2][Mpad][4] one (element j of a quad = input channel
                                     8 q + 2 j + parity) that the 16-byte-fragment kernels read */
    int32_t split;                /* nonzero (needs packed_v3): opt-in split precision.  A third image of the same size follows,
                                     [tap][Cin_pad/16][hi|lo][h][Mpad] x 8 bf16 (input channel 16 q + 8 h + e; w = hi + lo, hi =
                                     bf16(w) round-to-nearest-even, lo = bf16(w - hi)); layers with >= 16 input channels per
                                     group and more than 16 output channels are then computed as hi*hi + hi*lo + lo*hi on the
                                     bf16 matrix pipe with fp32 accumulation (csrc/conv_ws3s.h: ~1e-5 relative to the fp32
                                     kernels); the other layers run the fp32 kernels unchanged.
                                     2 (no third image): fp16 operands -- the reference's is_half mode (src/rvc.py:103-104,137-138)
                                     -- where the layer takes csrc/conv_g1.h (1 x 1 GEMM) or csrc/conv_g1w.h (wino == 8): both
                                     operands rounded to fp16 (nearest even) in registers in front of v_mfma_f32_32x32x8_f16, fp32
                                     activations, accumulation and epilogue (~1e-3 relative to the fp32 kernels); layers on any
                                     other kernel run in fp32 unchanged */
    int32_t wino;                 /* nonzero: w_packed holds the packed images of the (Cout, Cin, 3, 4) Winograd F(2, 3) kernel of a 3 x 3,
                                     stride 1, dilation 1, padding 1, groups 1 layer -- U[.][.][kh][0..3] = (g0, (g0 + g1 + g2) / 2,
                                     (g0 - g1 + g2) / 2, g2) of row kh -- and the layer runs csrc/conv_ws3w.h (12 instead of 18
                                     contractions per output pair).  Needs packed_v3, no residual / accumulate / shuffle /
                                     pre-activation, act none or ReLU, out_scale 1.
                                     2: the two-dimensional form F(2 x 2, 3 x 3) (csrc/conv_w2d.h: 16 instead of 36 contractions per
                                     2 x 2 output block) of the same kind of layer; w_packed is ONE image,
                                     [Cout / 48][ceil(Cin / 8)][s = 0..1][point p = 4 i + q][ks = 0..3][m = 0..47] floats with element
                                     U[48 mu + m][8 chunk + 4 s + ks][i][q], U = G g G^T, G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
                                     (zero beyond Cin).  Needs Cout % 48 == 0, W % 4 == 0, x 16-byte aligned with strides % 4 == 0.
                                     12: the same kernel reading PAIR fragments -- the image with the two points of a pair side by side,
                                     [Cout / 48][ceil(Cin / 8)][s][p / 2][ks][m][p % 2]: one 8-byte LDS read per two MFMAs (round 6: 1-6 %
                                     faster on the MDX-Net levels; what aicovergen_amd routes by default).  (3: four waves per
                                     workgroup; 4 / 5: eight / four waves on QUAD fragments [s][p / 4][ks][m][p % 4].)
                                     8: the ONE-dimensional form F(2, 3) of a k = 3 / 7 / 11, stride 1, dilation 1, "same"-padded 1-D layer
                                     (csrc/conv_g1w.h: 4 / 10 / 15 instead of 6 / 14 / 22 contractions per output pair -- the vocoder's
                                     ResBlocks, src/infer_pack/modules.py:299-312); w_packed holds the packed images of the
                                     (Cout, Cin, 1, S) SLOT kernel: per 3-tap group (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2), a
                                     remainder of two taps (a, a + b, b), of one tap (a, -a).  Residual / accumulate / out_scale and a
                                     leaky-ReLU input activation are supported; needs W % 4 == 0 and 16-byte aligned rows of x, y, res */
    int32_t gemm_tile;            /* 1 x 1 layers the LDS-DMA staged GEMM can take (csrc/conv_g1.h: unit stride, no padding, one group, no
                                     input activation but a leaky ReLU, contiguous 16-byte-aligned maps of a multiple of 4 positions): 0 the library's
                                     policy; 1 never that kernel; 2 / 3 / 4 its 128 x 256 / 64 x 256 / 192 x 256 tile (rows x positions
                                     per workgroup).  Layers that kernel cannot take ignore the field.  (Development builds only: 12 / 13 force
                                     the k-tap 1-D variant csrc/conv_g1k.h -- measured slower than the default kernels on the vocoder's
                                     layers, not in the product library) */
} aicg_conv_desc;

int aicg_conv_bkc(int taps);
int aicg_conv_desc_size(void);
int aicg_conv_forward(const aicg_conv_desc* desc, const float* x, const float* w_packed, const float* bias,
                      const float* res, float* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Transposed convolution, second half: cols = (Cout*KH*KW, Cin) x input computed by aicg_conv_forward as a
 * 1x1 GEMM, then gathered ("col2im") here:  torch.nn.ConvTranspose1d of GeneratorNSF.ups
 * (src/infer_pack/models.py:453-463, :503) and ConvTranspose2d of rmvpe.ResDecoderBlock (src/rmvpe.py:147-155).
 *   out[n,co,ho,wo] = act(bias[co] + sum_{kh,kw} cols[n,(co*KH+kh)*KW+kw,hi,wi]) + add[n,co,ho,wo],
 *   hi = (ho + pad_h - kh)/stride_h when divisible and in range (same for w).  add may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int aicg_col2im(const float* cols, const float* bias, const float* add, float* out, int N, int Cout, int Hi,
                int Wi, int Ho, int Wo, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w,
                int act, float slope, int64_t o_sn, int64_t o_sc, int64_t o_sh, int64_t a_sn, int64_t a_sc,
                int64_t a_sh, void* stream);

/* NSF harmonic source: SineGen.forward + SourceModuleHnNSF.forward (src/infer_pack/models.py:320-370,414-419)
 * with harmonic_num = 0.  f0: [T] Hz per frame; noise: [T*upp] standard-normal draws (the reference's
 * torch.randn_like at :368, supplied by the caller so that both sides of a parity test see the same values);
 * prefix_scratch: [T] doubles; out: [T*upp] = tanh(lin_w * (sine*uv + noise_amp*noise) + lin_b). */
int aicg_sine_source(const float* f0, const float* noise, double* prefix_scratch, float* out, int T, int upp,
                     float sr, float sine_amp, float noise_std, float lin_w, float lin_b, void* stream);

/* commons.fused_add_tanh_sigmoid_multiply (src/infer_pack/commons.py:105-112); the conditioning slice is a
 * per-channel constant for a (1,C,1) speaker embedding and is folded into the producing conv's bias.
 * a: (N, 2C, T) -> out: (N, C, T). */
int aicg_gate_tanh_sigmoid(const float* a, float* out, int N, int C, int64_t T, void* stream);

/* z_p = m_p + exp(logs_p) * noise * scale (src/infer_pack/models.py:748); stats = [m_p ; logs_p] (2C, T). */
int aicg_prior_sample(const float* stats, const float* noise, float* out, int C, int64_t T, float scale, void* stream);

/* VC.vc feature plumbing (src/vc_infer_pipeline.py:433-452): nearest x2 upsample of the (Th, C) HuBERT
 * features, protect blend with the voiced mask, written channel-major (C, T).  feats0/pitchf NULL = no protect. */
int aicg_feats_prepare(const float* feats, const float* feats0, const float* pitchf, float* out, int Th, int C,
                       int T, float protect, void* stream);

/* modules.LayerNorm over channels of (N, C, T) maps with the residual add of attentions.Encoder fused
 * (src/infer_pack/modules.py:29-32, attentions.py:67,71): out = LN_c(x + res) * gamma + beta. res may be NULL. */
int aicg_layernorm_ct(const float* x, const float* res, const float* gamma, const float* beta, float* out, int N,
                      int C, int64_t T, float eps, int64_t x_sn, int64_t r_sn, int64_t o_sn, void* stream);

/* Per-row statistics over time + affine + activation: HuBERT feature extractor GroupNorm(512, 512) + GELU (fairseq
 * ConvFeatureExtractionModel layer 0; call site src/vc_infer_pipeline.py:398-406).  `workspace`: NULL, or
 * aicg_rownorm_act_workspace_floats(rows, T) floats of scratch -- with it, rows of >= 4096 elements take the split form (partial
 * moments per row segment, then one float4 normalise + activate stream); without it one workgroup per row. */
int aicg_rownorm_act_workspace_floats(int rows, int64_t T, int64_t* n_floats);
int aicg_rownorm_act(const float* x, const float* gamma, const float* beta, float* out, int rows, int64_t T,
                     float eps, int act, float* workspace, void* stream);
/* ... with a row stride `ld` >= T shared by x and out (ABI 4): the extractor keeps its activations in rows padded to a multiple of four
 * floats so that the stride-2 layers behind can be staged by 16-byte DMA (csrc/conv_g1s.h); the padding is neither read nor written */
int aicg_rownorm_act_ld(const float* x, const float* gamma, const float* beta, float* out, int rows, int64_t T, int64_t ld,
                        float eps, int act, float* workspace, void* stream);

/* Fused softmax attention over channel-major q/k/v (H*D, T): o = softmax(scale * q^T k + relk) v.
 * relk (H, 2*window+1, T) holds q_i . E^k_m (attentions.py:238-243) or is NULL (HuBERT).  lse (H, T) receives the
 * log-sum-exp per query (needed by aicg_attention_relv) or is NULL.  D in {32, 64, 96, 128}. */
int aicg_attention(const float* q, const float* k, const float* v, const float* relk, float* o, float* lse, int T,
                   int H, int D, int window, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                   void* stream);
/* Same result with the key range split over n_splits workgroup groups (1..16, at most ceil(T/32)) and a log-sum-exp
 * merge pass: fills the GPU when ceil(T/128)*H is small (enc_p: 2 heads).  scratch: n_splits * H * (D + 2) * T floats
 * (unused when n_splits == 1). */
int aicg_attention_split(const float* q, const float* k, const float* v, const float* relk, float* o, float* lse,
                         int T, int H, int D, int window, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                         float scale, int n_splits, float* scratch, void* stream);
/* o_i += sum_{|j-i|<=window} P_ij E^v_{j-i+window} (attentions.py:264-271); relv_emb: (2*window+1, D). */
int aicg_attention_relv(const float* q, const float* k, const float* relk, const float* relv_emb,
                        const float* lse, float* o, int T, int H, int D, int window, int64_t ldq, int64_t ldk,
                        int64_t ldo, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RMVPE f0 estimator helpers (reference src/rmvpe.py)
 * ---------------------------------------------------------------------------------------------- */
/* magnitude = sqrt(re^2 + im^2) (src/rmvpe.py:314) */
int aicg_complex_abs(const float* re, const float* im, float* out, int64_t n, void* stream);
/* eval-mode BatchNorm2d on the network input as a per-channel affine (src/rmvpe.py:74,92) */
int aicg_channel_affine(const float* x, const float* scale, const float* shift, float* out, int N, int C,
                        int64_t HW, int act, void* stream);
/* nn.AvgPool2d(kernel_size=(2,2)) (src/rmvpe.py:111); x: (N,C,H,W) view with strides, out contiguous (N,C,H/2,W/2) */
int aicg_avgpool2x2(const float* x, float* out, int N, int C, int H, int W, int64_t x_sn, int64_t x_sc,
                    int64_t x_sh, void* stream);
/* Bidirectional single-layer nn.GRU recurrence (src/rmvpe.py:11-20).  gi: (2*3*hidden, T) input projections
 * W_ih x + b_ih for [forward r,z,n ; reverse r,z,n]; whh_t: (2, hidden, 3*hidden) = W_hh^T per direction;
 * bhh: (2*3*hidden); out: (2*hidden, T) = [forward h ; reverse h]. */
int aicg_gru_bidir(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T,
                   void* stream);
/* Same recurrence with each direction split over two co-resident workgroups that keep all of W_hh on chip and exchange
 * the new hidden state every step through tagged 8-byte granules (agent-scope stores / relaxed polls).
 * xchg_scratch: 32 * hidden + 64 bytes of device memory (zeroed by the call); the last int is set to 1 on a spin timeout. */
int aicg_gru_bidir_2wg(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T,
                       void* xchg_scratch, void* stream);
/* The same recurrence with each direction split over FOUR workgroups (hidden = 256 only): all recurrent weights in registers, half
 * the dot product per step, the three partners' quarters fetched in one poll round.  Same scratch size and error protocol as
 * aicg_gru_bidir_2wg (the exchange-timeout word is the first int behind the 32 * hidden granule bytes). */
int aicg_gru_bidir_4wg(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T,
                       void* xchg_scratch, void* stream);
/* The recurrence in SEGMENTS: steps s_begin .. s_end - 1 of the T (the forward direction visits frame s, the backward one frame
 * T - 1 - s), resuming from / leaving its hidden state in h_state (2, hidden) (may be NULL for a segment that starts at 0 and is not
 * continued).  Cut this way the recurrence is the same arithmetic step by step -- bit-identical to one call over 0 .. T -- but the
 * frames both directions have passed become available segment by segment: the multi-GPU pipeline starts synthesising the middle
 * chunks of a track while the recurrence (replicated on every rank) still runs towards its ends.  The segments of one
 * recurrence share xchg_scratch (cleared by the segment that starts at 0). */
int aicg_gru_bidir_seg(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T, int64_t s_begin,
                       int64_t s_end, float* h_state, void* stream);
int aicg_gru_bidir_4wg_seg(const float* gi, const float* whh_t, const float* bhh, float* out, int hidden, int64_t T, int64_t s_begin,
                           int64_t s_end, float* h_state, void* xchg_scratch, void* stream);
/* RMVPE.decode / to_local_average_cents (src/rmvpe.py:359-364,385-409).  salience: (T, n_bins) row-major fp32;
 * cents, f0: (T) float64 (bit-equal to the numpy reference given identical salience); center: (T) argmax or NULL. */
int aicg_salience_decode(const float* salience, double* cents, double* f0, int* center, int64_t T, int n_bins,
                         float thred, void* stream);
/* VC.get_f0 tail (src/vc_infer_pipeline.py:346,361-368): f0_out = f0_in * factor; coarse = rint(clamp(mel-scale
 * affine map to [1,255])) as int64 (np.rint, round-half-even). */
int aicg_f0_coarse(const double* f0_in, double factor, double* f0_out, int64_t* coarse, int64_t n, double mel_min,
                   double mel_max, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MDX-Net separator helpers.  The reference runs the network as an opaque ONNX graph (src/mdx.py:74-77,193);
 * its layers map onto aicg_conv_forward / aicg_col2im plus the last-axis Linear below.
 * ---------------------------------------------------------------------------------------------- */
/* c[r][o] = act((sum_i a[r][i] w[o][i] + bias[o]) * row_scale[ch(r)] + row_shift[ch(r)]) + res[r][o],
 * ch(r) = (r / rows_per_ch) % n_ch.  nn.Linear over the last (frequency) axis of a (B,C,T,F) map followed by
 * eval-mode BatchNorm2d(C) + ReLU (the TDF block), with the TFC_TDF residual add fused. */
int aicg_gemm_nt(const float* a, const float* w, const float* bias, const float* row_scale,
                 const float* row_shift, const float* res, float* c, int64_t R, int K, int O, int64_t lda,
                 int64_t ldw, int64_t ldc, int64_t ldr, int rows_per_ch, int n_ch, int act, void* stream);
/* the same contract in the opt-in split precision (both operands carried as bf16 hi + lo, products hi*hi + hi*lo + lo*hi on the
 * bf16 matrix pipe, fp32 accumulation and epilogue; ~4e-6 relative to aicg_gemm_nt -- see aicg_conv_desc.split) */
int aicg_gemm_nt_split(const float* a, const float* w, const float* bias, const float* row_scale,
                       const float* row_shift, const float* res, float* c, int64_t R, int K, int O, int64_t lda,
                       int64_t ldw, int64_t ldc, int64_t ldr, int rows_per_ch, int n_ch, int act, void* stream);

/* The whole TDF block of the MDX-Net U-Net in one launch (graph executed at src/mdx.py:74-77,193):
 *   out = x + relu(bn2(relu(bn1(x W1^T + b1)) W2^T + b2)),  x / out (R, F) contiguous rows of a (B, C, T, F) map, W1 (H, F), W2 (F, H).
 * w1_packed: W1 re-laid-out as [F / 8][2][H][4] with element e of quad (g, par, h) = W1[h][8 g + 2 e + par]; w2_packed: W2 as
 * F / 32 slabs of 32 rows x (H + 4) floats, each slab zero-padded to a multiple of 256 floats (the LDS images of the stages);
 * s*, t*: eval BatchNorm2d scale / shift of the channel a row belongs to, ch = (row / rows_per_ch) % n_ch (NULL: none).
 * The (R, H) intermediate stays in the accumulator registers (csrc/tdf_pair.hip).  aicg_tdf_pair_supported returns 1 for the
 * geometries instantiated (H = 32 x {2, 3, 4, 6, 8, 12}, F % 32 == 0, rows_per_ch % 32 == 0); others take two aicg_gemm_nt calls. */
int aicg_tdf_pair_supported(int F, int H, int rows_per_ch);
int aicg_tdf_pair(const float* x, const float* w1_packed, const float* b1, const float* s1, const float* t1, const float* w2_packed,
                  const float* b2, const float* s2, const float* t2, float* out, int64_t R, int F, int H, int rows_per_ch,
                  int n_ch, void* stream);
/* out = a * b elementwise (U-Net skip connection of the TFC-TDF net: x *= ds_outputs[-i-1]) */
int aicg_mul(const float* a, const float* b, float* out, int64_t n, void* stream);
/* run_mdx epilogue pieces (src/mdx.py:259-267,280): out = alpha * a + beta * b (+ gamma * c if c) */
int aicg_axpbypcz(const float* a, float alpha, const float* b, float beta, const float* c, float gamma, float* out,
                  int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * VC.pipeline pre/post-processing on the device (reference src/vc_infer_pipeline.py)
 * ---------------------------------------------------------------------------------------------- */
/* out[j] = sum_{i<window} x[j+i], j < n, summed in ascending i like the reference loop (:518-519); x has n+window-1 */
int aicg_box_sum_f64(const double* x, double* out, int64_t n, int window, void* stream);
/* per segment s: out[s] = first index of min |x[starts[s] + i]|, i < lens[s]  (np.where(a == a.min())[0][0], :524-527) */
int aicg_argmin_abs_f64(const double* x, const int64_t* starts, const int64_t* lens, int64_t* out, int n_seg,
                        void* stream);
/* librosa.feature.rms(y=x, frame_length, hop_length) with center=True / reflect padding (:43-46); x is float32 or
 * float64 (is_f64); out: 1 + n/hop_length float64 values */
int aicg_frame_rms(const void* x, int is_f64, double* out, int64_t n, int frame_length, int hop_length, void* stream);
/* data *= interp(rms1)^(1-rate) * max(interp(rms2), 1e-6)^(rate-1), linear interpolation to n points (:47-59) */
int aicg_rms_mix(float* data, int64_t n, const double* rms1, int64_t m1, const double* rms2, int64_t m2, double rate,
                 void* stream);
/* out[0] = max |x| (:645) */
int aicg_absmax(const float* x, int64_t n, float* out, void* stream);
/* out = (x * scale).astype(int16), truncating (:649) */
int aicg_to_int16(const float* x, int16_t* out, int64_t n, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CREPE f0 (torchcrepe 0.0.20 `predict`, called at src/vc_infer_pipeline.py:116-126 for f0_method mangio-crepe)
 * ---------------------------------------------------------------------------------------------- */
/* torchcrepe.preprocess: per 1024-sample frame, x = (x - mean) / max(1e-10, unbiased std) */
int aicg_frame_normalize(const float* x, float* out, int64_t n_frames, int frame_len, void* stream);
/* eval BatchNorm2d (per-channel scale/shift, applied AFTER the ReLU in CREPE) + MaxPool (2,1): (N,C,W) -> (N,C,W/2) */
int aicg_affine_maxpool2(const float* x, const float* scale, const float* shift, float* out, int N, int C, int W,
                         void* stream);
/* torchcrepe.decode.viterbi on each sequence independently: probs (n_seq, n_bins, max_steps) fp32 posteriors, bins
 * outside [bin_lo, bin_hi) masked to -inf, softmax over bins, librosa.sequence.viterbi with the 12-bin triangular
 * transition; bins_out (n_seq, max_steps) int64.  Scratch: logp (n_seq*max_steps*n_bins fp32), ptr (same, uint16). */
int aicg_crepe_viterbi(const float* probs, const int* seq_len, float* logp_scratch, uint16_t* ptr_scratch,
                       int64_t* bins_out, int n_seq, int n_bins, int max_steps, int bin_lo, int bin_hi, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Signal processing either side of the networks (csrc/dsp.hip)
 * ---------------------------------------------------------------------------------------------- */

/* 3-tap filter of a frame sequence over its in-range, non-NaN neighbours: mode 0 = lower median, 1 = mean (a mean of exactly 0 and
 * an empty window give NaN).  torchcrepe.filter.median(pd, 3) / .mean(f0, 3) of the 'crepe' / 'crepe-tiny' f0 methods
 * (src/vc_infer_pipeline.py:160-161). */
int aicg_filter3(const float* x, float* out, int64_t n, int mode, void* stream);

/* scipy.signal.filtfilt(b, a, x) with its defaults (odd extension by padlen samples, lfilter_zi initial conditions), float64:
 * the 48 Hz Butterworth high-pass of VC.pipeline (src/vc_infer_pipeline.py:22,513).  b, a (order + 1 entries) and zi (order
 * entries, scipy.signal.lfilter_zi) are HOST arrays; x, y (n samples) and the scratch ext, mid (n + 2 padlen samples each) are
 * device arrays.  The recurrence runs block-parallel: each thread re-runs `warm` samples in front of its `block` outputs from a
 * zero state (the caller picks warm so that max|pole|^warm is below float64 rounding). */
int aicg_filtfilt_f64(const double* x, double* y, int64_t n, const double* b, const double* a, const double* zi, int order,
                      int padlen, int block, int warm, double* ext, double* mid, void* stream);

/* y[i] = sum_m h[(i + pre) * down - m * up] * mean_c x[c][m]: scipy.signal.resample_poly / upfirdn with the filter given as the
 * polyphase table hp[phase][tap] = h[phase + tap * up] (taps per phase, zero padded), plus the channel mean of a
 * (n_channels, n_in) input with channel stride x_sc.  The opt-in device hand-over of the separated vocals to the RVC stage
 * (44.1 kHz stereo -> 16 kHz mono), which the reference does through a PCM-16 WAV and ffmpeg (src/mdx.py:273,280,
 * src/my_utils.py:14-16). */
int aicg_resample_poly(const float* x, float* y, int64_t n_in, int64_t n_out, int n_channels, int64_t x_sc, int up, int down,
                       const float* hp, int taps, int64_t pre, void* stream);

/* Retrieval mix of VC.vc (src/vc_infer_pipeline.py:409-431: index.search(npy, k=8), inverse-square weights, blend; the index is
 * the faiss IVF-Flat file read at :497-512):
 * aicg_row_sqnorm: |v_r|^2 of a (rows, dim) matrix;
 * aicg_knn8: the 8 nearest columns by squared L2 distance |q|^2 - 2 q.x + |x|^2 from inner products dots[r][c] (row stride ld) of
 *   one column chunk [col_off, col_off + cols), merged into best_d / best_i (rows x 8, ascending, ties by lower column) when
 *   merge != 0; fewer than 8 columns in total leave distance +inf / index INT64_MAX.  Serves the coarse quantizer (columns =
 *   centroids: faiss IndexFlatL2's BLAS search uses the same expansion) and the opt-in exhaustive search (columns = all vectors);
 * aicg_ivf_scan8: IndexIVFFlat::search -- query r scans inverted lists probe[r][0..nprobe) (row stride probe_ld) of vectors
 *   stored list by list (list l = vecs[list_off[l] .. list_off[l + 1])), distances computed directly as sum (q - x)^2, the 8
 *   smallest kept (a later candidate must be strictly closer: faiss' heap rule); best_pos = position in `vecs`, -1 / +inf when
 *   fewer than 8 vectors were visited;
 * aicg_index_mix: feats = rate * sum_k w_k big[best_i[k]] + (1 - rate) * feats with w = (1/d)^2 / sum (1/d)^2; best_i < 0 gets
 *   weight 0; exact-zero distances share the whole weight (the reference divides inf by inf there); recompute != 0 re-evaluates
 *   d_k = sum (feats - big[best_i[k]])^2 first and stores it in best_d. */
int aicg_row_sqnorm(const float* v, float* out, int64_t rows, int dim, void* stream);
int aicg_knn8(const float* dots, int64_t ld, const float* xnorm, const float* qnorm, int rows, int cols, int64_t col_off,
              float* best_d, int64_t* best_i, int merge, void* stream);
int aicg_ivf_scan8(const float* q, const float* vecs, const int64_t* list_off, const int64_t* probe, int probe_ld, int nprobe,
                   int rows, int dim, float* best_d, int64_t* best_pos, void* stream);
int aicg_index_mix(float* feats, const float* big, float* best_d, const int64_t* best_i, int rows, int dim, float rate,
                   int recompute, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AICG_H_ */
