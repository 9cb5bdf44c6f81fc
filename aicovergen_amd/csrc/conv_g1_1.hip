// instantiation unit: LDS-DMA staged 1 x 1 GEMM (conv_g1.h)
#include "conv_g1.h"
namespace aicg {
int run_g1_128x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1<2, 2, 2, 2>(p, st); }
int run_g1_64x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1<1, 2, 2, 3>(p, st); }
int run_g1_192x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1<3, 2, 2, 1>(p, st); }
#ifdef AICG_DEV_SWITCHES
int run_g1_256x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1<4, 2, 2, 1>(p, st); }
int run_g1_128x512(ConvArgs& p, hipStream_t st) { return launch_conv_g1<4, 1, 4, 1>(p, st); }
int run_g1_burst(ConvArgs& p, hipStream_t st, int code) {
    return code == 2 ? launch_conv_g1<2, 2, 2, 2, false>(p, st) : code == 3 ? launch_conv_g1<1, 2, 2, 3, false>(p, st) : launch_conv_g1<3, 2, 2, 1, false>(p, st);
}
#endif
}  // namespace aicg
