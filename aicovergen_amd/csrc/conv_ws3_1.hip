// 16-byte-fragment conv tiles 128x128 and 96x128 (kernel templates: conv_ws3.h)
#include "conv_ws3.h"

namespace aicg {
int run_ws3_128x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3<128, 128, 1, 4, 32>(p, st); }
int run_ws3_96x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3<96, 128, 1, 4, 64>(p, st); }
}  // namespace aicg
