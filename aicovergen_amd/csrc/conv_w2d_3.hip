// dev library only (-DAICG_CONV_ABLATION): profiling variants of the routed form of conv_w2d.h (eight waves, pair fragments, tied MFMAs)
#include "conv_w2d.h"
namespace aicg {
#ifdef AICG_DEV_SWITCHES
int run_w2d_pairs_ablation(ConvArgs& p, hipStream_t st, int bits) {
    switch (bits) {
        case 16384: return launch_conv_w2d<8, 2, 16384>(p, st);    // MFMAs through the builtin (rounds 4-6: accumulator quads through scratch memory)
        case 65536: return launch_conv_w2d<8, 2, 65536>(p, st);    // the epilogue one output channel at a time (rounds 4-6)
#ifdef AICG_CONV_ABLATION
        case 262144: return launch_conv_w2d<8, 2, 262144>(p, st);  // the upper wave of a SIMD at issue priority 1
        case 262144 | 256: return launch_conv_w2d<8, 2, 262144 | 256>(p, st);
        case 8192: return launch_conv_w2d<8, 2, 8192>(p, st);      // place() at the top of the stage that needs it
        case 32768: return launch_conv_w2d<8, 2, 32768>(p, st);    // no stage-end DMA wait (races: wrong results)
        case 64: return launch_conv_w2d<8, 2, 64>(p, st);
        case 256: return launch_conv_w2d<8, 2, 256>(p, st);
        case 16: return launch_conv_w2d<8, 2, 16>(p, st);
        case 128: return launch_conv_w2d<8, 2, 128>(p, st);
        case 1: return launch_conv_w2d<8, 2, 1>(p, st);
        case 2: return launch_conv_w2d<8, 2, 2>(p, st);
        case 4: return launch_conv_w2d<8, 2, 4>(p, st);
        case 8: return launch_conv_w2d<8, 2, 8>(p, st);
        case 32: return launch_conv_w2d<8, 2, 32>(p, st);
        case 1 | 16: return launch_conv_w2d<8, 2, 1 | 16>(p, st);
        case 2 | 4: return launch_conv_w2d<8, 2, 2 | 4>(p, st);
        case 1 | 4 | 16 | 32: return launch_conv_w2d<8, 2, 1 | 4 | 16 | 32>(p, st);
        case 1 | 2 | 4 | 16 | 32: return launch_conv_w2d<8, 2, 1 | 2 | 4 | 16 | 32>(p, st);
#endif
        default: return 1;
    }
}
#else
int run_w2d_pairs_ablation(ConvArgs&, hipStream_t, int) { return 1; }
#endif
}  // namespace aicg
