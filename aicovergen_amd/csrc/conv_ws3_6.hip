// 16x16x4 conv tiles on 8-byte fragments: 48 and 16 rows x 256 positions (kernel templates: conv_ws3.h)
#include "conv_ws3.h"

namespace aicg {
int run_ws3m16h_48(ConvArgs& p, hipStream_t st) { return launch_conv_ws3m16h<48>(p, st); }
int run_ws3m16h_16(ConvArgs& p, hipStream_t st) { return launch_conv_ws3m16h<16>(p, st); }
}  // namespace aicg
