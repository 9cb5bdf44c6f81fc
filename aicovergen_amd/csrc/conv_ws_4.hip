// Wave-specialised conv tiles with four MFMAs per consumer k-step for 128- / 64- / 32-row layers (kernel templates: conv_kernels.h)
// development builds only (AICG_DEV_SWITCHES): no default dispatch path selects these kernels
#ifdef AICG_DEV_SWITCHES
#include "conv_kernels.h"

namespace aicg {
int run_ws_128x128_k32(ConvArgs& p, hipStream_t st) { return launch_conv_ws<128, 128, 1, 4, 32>(p, st); }
int run_ws_64x256(ConvArgs& p, hipStream_t st) { return launch_conv_ws<64, 256, 1, 4, 64>(p, st); }
int run_ws_32x512(ConvArgs& p, hipStream_t st) { return launch_conv_ws<32, 512, 1, 4, 64>(p, st); }
}  // namespace aicg
#endif
