// fp32 instantiations of the tile convolution (conv_tile.h); a translation unit of its own for build time only.
#include "conv_tile.h"
namespace hrp { int launch_conv_f32(const hrp_conv_desc& d, hipStream_t s) { return launch_conv<float>(d, s); } }
