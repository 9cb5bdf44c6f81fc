// Wave-specialised conv tiles 96x128 and 64x128 (kernel templates: conv_kernels.h)
#include "conv_kernels.h"

namespace aicg {
int run_ws_96x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws<96, 128, 1, 4, 64>(p, st); }
int run_ws_64x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws<64, 128, 2, 2, 64>(p, st); }
}  // namespace aicg
