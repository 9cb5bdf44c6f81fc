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

const char* aicg_last_error(void);
/* ABI version; bumped whenever a signature changes */
int aicg_abi_version(void);

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

#ifdef __cplusplus
}
#endif
#endif /* AICG_H_ */
