// bf16 instantiations of the batched tile convolution (conv_batch.h).
#include "conv_batch.h"
namespace hrp {
int conv_batch_prepare_bf16(const hrp_conv_desc* descs, int n, void* table, hrp_batch_info* info) { return conv_batch_prepare_t<bf16_t>(descs, n, table, info); }
int conv_batch_launch_bf16(const void* table_dev, const hrp_batch_info* info, hipStream_t s) { return conv_batch_launch_t<bf16_t>(table_dev, info, s); }
int64_t conv_batch_table_bytes(int n) { return (int64_t)n * sizeof(ConvProblem); }
}

#ifdef HRP_TIMELINE
// (the batched kernels of this translation unit stamp their own copy of the timeline array)
extern "C" int hrp_debug_conv_timeline_batch(void* dst, int nblocks, int clear) {
  if (dst) (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hrp::g_conv_timeline), sizeof(unsigned long long) * 8 * nblocks);
  if (clear) {
    void* p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(hrp::g_conv_timeline));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 8);
  }
  return 0;
}
#endif
