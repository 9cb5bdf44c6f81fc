// instantiation unit: LDS-DMA staged stride-2 k-tap 1-D convolution (conv_g1s.h)
#include "conv_g1s.h"
namespace aicg {
int run_g1s_128x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1s<2, 2, 2, 2>(p, st); }
int run_g1s_64x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1s<1, 2, 2, 3>(p, st); }
}  // namespace aicg
