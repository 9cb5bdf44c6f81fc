// instantiation unit: conv_g1w.h, the persistent tile walk (dilation 1; development builds: measured 2-5 % slower than one tile per workgroup)
#ifdef AICG_DEV_SWITCHES
#include "conv_g1w.h"
namespace aicg {
int run_g1w_32x512_pers(ConvArgs& p, hipStream_t st) { return launch_conv_g1w<1, 4, 2, 0, false, true>(p, st); }
}  // namespace aicg
#endif
