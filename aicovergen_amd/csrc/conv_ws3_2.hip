// 16-byte-fragment conv tiles 64x128 and 64x64 (kernel templates: conv_ws3.h)
#include "conv_ws3.h"

namespace aicg {
int run_ws3_64x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3<64, 128, 2, 2, 64>(p, st); }
int run_ws3_64x64(ConvArgs& p, hipStream_t st) { return launch_conv_ws3<64, 64, 2, 2, 64>(p, st); }
}  // namespace aicg
