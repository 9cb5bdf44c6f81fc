// 16-byte-fragment conv tiles 32x256 and 32x128 (kernel templates: conv_ws3.h)
#include "conv_ws3.h"

namespace aicg {
int run_ws3_32x256(ConvArgs& p, hipStream_t st) { return launch_conv_ws3<32, 256, 1, 4, 64>(p, st); }
int run_ws3_32x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3<32, 128, 1, 4, 64>(p, st); }
}  // namespace aicg
