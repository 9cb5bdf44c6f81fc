// Split-precision (bf16 x 3) conv tiles 128x128 and 96x128 (kernel templates: conv_ws3s.h)
#include "conv_ws3s.h"

namespace aicg {
int run_ws3s_128x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3s<128, 128, 1, 4, 32>(p, st); }
int run_ws3s_96x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3s<96, 128, 1, 4, 64>(p, st); }
}  // namespace aicg
