// explicit instantiations: Winograd F(2,3)-along-rows 3 x 3 convolution (conv_ws3w.h)
#include "conv_ws3w.h"
namespace aicg {
int run_ws3w_96(ConvArgs& p, hipStream_t st) { return launch_conv_ws3w<96, 4>(p, st); }
int run_ws3w_64(ConvArgs& p, hipStream_t st) { return launch_conv_ws3w<64, 4>(p, st); }
int run_ws3w_48(ConvArgs& p, hipStream_t st) { return launch_conv_ws3w<48, 8>(p, st); }
int run_ws3w_32(ConvArgs& p, hipStream_t st) { return launch_conv_ws3w<32, 4>(p, st); }
}  // namespace aicg
