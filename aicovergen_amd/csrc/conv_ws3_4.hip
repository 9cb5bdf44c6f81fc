// 16-byte-fragment conv tiles on the 16x16x4 MFMA for 48- and 16-row layers (kernel templates: conv_ws3.h)
#include "conv_ws3.h"

namespace aicg {
int run_ws3m16_48(ConvArgs& p, hipStream_t st) { return launch_conv_ws3m16<48>(p, st); }
int run_ws3m16_16(ConvArgs& p, hipStream_t st) { return launch_conv_ws3m16<16>(p, st); }
}  // namespace aicg
