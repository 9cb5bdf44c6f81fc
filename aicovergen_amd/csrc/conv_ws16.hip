// 16x16x4-MFMA conv tiles for 48- and 16-row layers (kernel templates: conv_kernels.h)
#include "conv_kernels.h"

namespace aicg {
int run_m16_48(ConvArgs& p, hipStream_t st) { return launch_conv16<48>(p, st); }
int run_m16_16(ConvArgs& p, hipStream_t st) { return launch_conv16<16>(p, st); }
}  // namespace aicg
