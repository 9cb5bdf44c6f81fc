// instantiation unit: conv_g1w.h, the A/B variants (dilation 1 only; development builds): 64 x 256 tile, explicit MFMA / VALU interleave
#ifdef AICG_DEV_SWITCHES
#include "conv_g1w.h"
namespace aicg {
int run_g1w_64x256(ConvArgs& p, hipStream_t st) { return launch_conv_g1w<2, 2, 2, 0, false>(p, st); }
int run_g1w_32x512_sched(ConvArgs& p, hipStream_t st) { return launch_conv_g1w<1, 4, 2, 1, false>(p, st); }
}  // namespace aicg
#endif
