// instantiation unit: Winograd F(2 x 2, 3 x 3) 3 x 3 convolution (conv_w2d.h)
#include "conv_w2d.h"
namespace aicg {
int run_w2d_8(ConvArgs& p, hipStream_t st) { return launch_conv_w2d<8, 1>(p, st); }
int run_w2d_4(ConvArgs& p, hipStream_t st) { return launch_conv_w2d<4, 3>(p, st); }
int run_w2d_8q(ConvArgs& p, hipStream_t st) { return launch_conv_w2d<8, 0>(p, st); }
int run_w2d_8p(ConvArgs& p, hipStream_t st) { return launch_conv_w2d<8, 2>(p, st); }
#ifdef AICG_DEV_SWITCHES
int run_w2d_4p2(ConvArgs& p, hipStream_t st) { return launch_conv_w2d<4, 2, 0, 1>(p, st); }   // two workgroups per CU: measured level with the eight-wave form
#else
int run_w2d_4p2(ConvArgs&, hipStream_t) { return 1; }
#endif
int run_w2d_4q(ConvArgs& p, hipStream_t st) { return launch_conv_w2d<4, 0>(p, st); }
}  // namespace aicg
