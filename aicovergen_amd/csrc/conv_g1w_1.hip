// instantiation unit: 1-D Winograd F(2, 3) k-tap convolution on LDS-DMA staged operands (conv_g1w.h), dilations 1 / 3 / 5
#include "conv_g1w.h"
namespace aicg {
int run_g1w_32x512(ConvArgs& p, hipStream_t st) { return launch_conv_g1w<1, 4, 2>(p, st); }
}  // namespace aicg
