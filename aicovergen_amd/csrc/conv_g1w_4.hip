// instantiation unit: 1-D Winograd F(2, 3) k-tap convolution (conv_g1w.h) on the fp16 matrix pipe -- aicg_conv_desc.split == 2, the reference's is_half mode
#include "conv_g1w.h"
namespace aicg {
int run_g1w_32x512_h(ConvArgs& p, hipStream_t st) { return launch_conv_g1w<1, 4, 2, 0, true, false, true>(p, st); }
}  // namespace aicg
