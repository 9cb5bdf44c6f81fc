// Single-role conv kernels, small tiles (kernel templates: conv_kernels.h)
#include "conv_kernels.h"

namespace aicg {
int run_sr_64x128(ConvArgs& p, hipStream_t st) { return launch_conv<64, 128, 2, 2>(p, st); }
int run_sr_64x64(ConvArgs& p, hipStream_t st) { return launch_conv<64, 64, 2, 2>(p, st); }
int run_sr_32x256(ConvArgs& p, hipStream_t st) { return launch_conv<32, 256, 1, 4>(p, st); }
int run_sr_32x128(ConvArgs& p, hipStream_t st) { return launch_conv<32, 128, 1, 4>(p, st); }
}  // namespace aicg
