// fp32 instantiations of the batched tile convolution (conv_batch.h).
#include "conv_batch.h"
namespace hrp {
int conv_batch_prepare_f32(const hrp_conv_desc* descs, int n, void* table, hrp_batch_info* info) { return conv_batch_prepare_t<float>(descs, n, table, info); }
int conv_batch_launch_f32(const void* table_dev, const hrp_batch_info* info, hipStream_t s) { return conv_batch_launch_t<float>(table_dev, info, s); }
}
