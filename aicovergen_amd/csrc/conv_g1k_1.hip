// instantiation unit: LDS-DMA staged k-tap 1-D convolution (conv_g1k.h) -- development builds only (see conv.hip)
#ifdef AICG_DEV_SWITCHES
#include "conv_g1k.h"
namespace aicg {
int run_g1k_128x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1k<2, 2, 2, 2>(p, st); }
int run_g1k_64x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1k<1, 2, 2, 3>(p, st); }
}  // namespace aicg
#endif
