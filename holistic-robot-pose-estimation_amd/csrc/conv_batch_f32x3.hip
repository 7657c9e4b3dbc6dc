// HRP_F32X3 instantiations of the batched tile convolution (conv_batch.h).
#include "conv_batch.h"
namespace hrp {
int conv_batch_prepare_f32x3(const hrp_conv_desc* descs, int n, void* table, hrp_batch_info* info) { return conv_batch_prepare_t<f32x3_t>(descs, n, table, info); }
int conv_batch_launch_f32x3(const void* table_dev, const hrp_batch_info* info, hipStream_t s) { return conv_batch_launch_t<f32x3_t>(table_dev, info, s); }
}
