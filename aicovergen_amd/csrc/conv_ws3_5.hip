// 16-byte-fragment conv tile 160x128 (kernel templates: conv_ws3.h)
#include "conv_ws3.h"

namespace aicg {
int run_ws3_160x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3<160, 128, 1, 4, 32>(p, st); }
}  // namespace aicg
