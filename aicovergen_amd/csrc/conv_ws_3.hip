// Wave-specialised conv tiles 64x64, 32x256 and 32x128 (kernel templates: conv_kernels.h)
#include "conv_kernels.h"

namespace aicg {
int run_ws_64x64(ConvArgs& p, hipStream_t st) { return launch_conv_ws<64, 64, 2, 2, 64>(p, st); }
int run_ws_32x256(ConvArgs& p, hipStream_t st) { return launch_conv_ws<32, 256, 1, 4, 64>(p, st); }
int run_ws_32x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws<32, 128, 1, 4, 64>(p, st); }
}  // namespace aicg
