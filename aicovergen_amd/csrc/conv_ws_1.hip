// Wave-specialised conv tiles 160x128 and 128x128 (kernel templates: conv_kernels.h)
#include "conv_kernels.h"

namespace aicg {
int run_ws_160x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws<160, 128, 1, 4, 32>(p, st); }
int run_ws_128x128(ConvArgs& p, hipStream_t st) { return launch_conv_ws<128, 128, 2, 4, 64>(p, st); }
}  // namespace aicg
