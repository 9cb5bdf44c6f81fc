// instantiation unit: 1-D Winograd F(2, 3) k-tap convolution on LDS-DMA staged operands (conv_g1w.h)
#include "conv_g1w.h"
namespace aicg {
int run_g1w_64x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1w<2, 2, 2>(p, st); }
int run_g1w_32x512(ConvArgs& p, hipStream_t st) { return launch_conv_g1w<1, 4, 2>(p, st); }
int run_g1w_32x512_sched(ConvArgs& p, hipStream_t st) { return launch_conv_g1w<1, 4, 2, 1>(p, st); }
}  // namespace aicg
