// dev library only (-DAICG_CONV_ABLATION): profiling variants of the eight-wave F(2 x 2, 3 x 3) kernel (conv_w2d.h, template parameter ABL)
#include "conv_w2d.h"
namespace aicg {
#ifdef AICG_CONV_ABLATION
int run_w2d_ablation(ConvArgs& p, hipStream_t st, int bits) {
    switch (bits) {
        case 64: return launch_conv_w2d<8, 1, 64>(p, st);
        case 16: return launch_conv_w2d<8, 1, 16>(p, st);
        case 128: return launch_conv_w2d<8, 1, 128>(p, st);
        case 1: return launch_conv_w2d<8, 1, 1>(p, st);
        case 2: return launch_conv_w2d<8, 1, 2>(p, st);
        case 4: return launch_conv_w2d<8, 1, 4>(p, st);
        case 8: return launch_conv_w2d<8, 1, 8>(p, st);
        case 32: return launch_conv_w2d<8, 1, 32>(p, st);
        case 1 | 4 | 16 | 32: return launch_conv_w2d<8, 1, 1 | 4 | 16 | 32>(p, st);
        case 1 | 2 | 4 | 16 | 32: return launch_conv_w2d<8, 1, 1 | 2 | 4 | 16 | 32>(p, st);
        case 1 | 2 | 4 | 16 | 32 | 64: return launch_conv_w2d<8, 1, 1 | 2 | 4 | 16 | 32 | 64>(p, st);
        case 2 | 4 | 8 | 16: return launch_conv_w2d<8, 1, 2 | 4 | 8 | 16>(p, st);
        case 1 | 16: return launch_conv_w2d<8, 1, 1 | 16>(p, st);
        case 2 | 4: return launch_conv_w2d<8, 1, 2 | 4>(p, st);
        case 256: return launch_conv_w2d<8, 1, 256>(p, st);
        case 512: return launch_conv_w2d<8, 1, 512>(p, st);
        case 1024: return launch_conv_w2d<8, 1, 1024>(p, st);
        case 2048: return launch_conv_w2d<8, 1, 2048>(p, st);
        default: return 1;
    }
}
#else
int run_w2d_ablation(ConvArgs&, hipStream_t, int) { return 1; }
#endif
}  // namespace aicg
