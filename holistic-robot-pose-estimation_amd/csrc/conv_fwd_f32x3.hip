// fp32-tensor / 3 x bf16-product instantiations of the tile convolution (conv_tile.h, HRP_F32X3); a translation unit of its own for
// build time only.
#include "conv_tile.h"
namespace hrp { int launch_conv_f32x3(const hrp_conv_desc& d, hipStream_t s) { return launch_conv<f32x3_t>(d, s); } }
