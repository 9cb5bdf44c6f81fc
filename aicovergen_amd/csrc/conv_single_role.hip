// Single-role conv kernels (layers the wave-specialised form cannot stage), large tiles (kernel templates: conv_kernels.h)
#include "conv_kernels.h"

namespace aicg {
int run_sr_160x128(ConvArgs& p, hipStream_t st) { return launch_conv<160, 128, 1, 4>(p, st); }
int run_sr_128x128_8w(ConvArgs& p, hipStream_t st) { return launch_conv<128, 128, 2, 4>(p, st); }
#ifdef AICG_DEV_SWITCHES
int run_sr_128x128_4w(ConvArgs& p, hipStream_t st) { return launch_conv<128, 128, 2, 2>(p, st); }
#endif
int run_sr_96x128(ConvArgs& p, hipStream_t st) { return launch_conv<96, 128, 1, 4>(p, st); }
}  // namespace aicg
