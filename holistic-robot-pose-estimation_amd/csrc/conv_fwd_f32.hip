// fp32 instantiations of the tile convolution (second translation unit of conv_fwd.hip, for build time only).
#define HRP_CONV_TU_F32
#include "conv_fwd.hip"
