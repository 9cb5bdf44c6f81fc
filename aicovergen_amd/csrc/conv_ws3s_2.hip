// Split-precision (bf16 x 3) conv tiles 64x128, 64x64, 32x256, 32x128 (kernel templates: conv_ws3s.h)
#include "conv_ws3s.h"

namespace aicg {
int run_ws3s_64x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3s<64, 128, 2, 2, 64>(p, st); }
int run_ws3s_64x64(ConvArgs& p, hipStream_t st) { return launch_conv_ws3s<64, 64, 2, 2, 64>(p, st); }
int run_ws3s_32x256(ConvArgs& p, hipStream_t st) { return launch_conv_ws3s<32, 256, 1, 4, 64>(p, st); }
int run_ws3s_32x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws3s<32, 128, 1, 4, 64>(p, st); }
int run_ws3s_64x256(ConvArgs& p, hipStream_t st) { return launch_conv_ws3s<64, 256, 1, 4, 32>(p, st); }
}  // namespace aicg
