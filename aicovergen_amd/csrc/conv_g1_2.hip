// instantiation unit: LDS-DMA staged 1 x 1 GEMM (conv_g1.h) on the fp16 matrix pipe -- aicg_conv_desc.split == 2, the reference's is_half mode
#include "conv_g1.h"
namespace aicg {
int run_g1_128x256_h(ConvArgs& p, hipStream_t st) { return launch_conv_g1<2, 2, 2, 2, true, true>(p, st); }
int run_g1_64x256_h(ConvArgs& p, hipStream_t st) { return launch_conv_g1<1, 2, 2, 3, true, true>(p, st); }
}  // namespace aicg
