// 16-byte-fragment conv tiles on the 16x16x4 MFMA for 48- and 16-row layers (kernel templates: conv_ws3.h)
// development builds only (AICG_DEV_SWITCHES): no default dispatch path selects these kernels
#ifdef AICG_DEV_SWITCHES
#include "conv_ws3.h"

namespace aicg {
int run_ws3m16_48(ConvArgs& p, hipStream_t st) { return launch_conv_ws3m16<48>(p, st); }
int run_ws3m16_16(ConvArgs& p, hipStream_t st) { return launch_conv_ws3m16<16>(p, st); }
}  // namespace aicg
#endif
