// Batched launches (include/hrp.h, hrp_batch_*): shared device / host plumbing.
//
// A batched kernel gets the launch table (device memory, one Problem struct per problem: descriptor + the tiling
// the host chose) and, by value, the first block of every problem.  A workgroup finds its problem with a scalar
// scan of that array (kernel arguments: SGPR loads, ~64 SALU cycles) and then runs the family's ordinary tile
// program on (descriptor, tiling, block index inside the problem).
#pragma once
#include "hrp_common.h"

namespace hrp {

struct BatchHdr {
  int blk0[HRP_BATCH_MAX];   // ascending; entries past the last problem hold INT_MAX
};

// -> problem index of block b; base = its first block
__device__ __forceinline__ int batch_find(const BatchHdr& h, int b, int& base) {
  int g = 0;
  base = 0;
#pragma unroll
  for (int i = 1; i < HRP_BATCH_MAX; ++i) {
    const bool ge = b >= h.blk0[i];
    g = ge ? i : g;
    base = ge ? h.blk0[i] : base;
  }
  return g;
}

static inline BatchHdr make_hdr(const int* blk0, int n) {
  BatchHdr h;
  for (int i = 0; i < HRP_BATCH_MAX; ++i) h.blk0[i] = i < n ? blk0[i] : 0x7fffffff;
  return h;
}

// family entry points (one translation unit each; batch.hip dispatches)
int conv_batch_prepare_bf16(const hrp_conv_desc* descs, int n, void* table, hrp_batch_info* info);
int conv_batch_prepare_f32(const hrp_conv_desc* descs, int n, void* table, hrp_batch_info* info);
int conv_batch_prepare_f32x3(const hrp_conv_desc* descs, int n, void* table, hrp_batch_info* info);
int conv_batch_launch_bf16(const void* table_dev, const hrp_batch_info* info, hipStream_t s);
int conv_batch_launch_f32(const void* table_dev, const hrp_batch_info* info, hipStream_t s);
int conv_batch_launch_f32x3(const void* table_dev, const hrp_batch_info* info, hipStream_t s);
int64_t conv_batch_table_bytes(int n);
int conv_check(const hrp_conv_desc* d);

int wgrad_batch_prepare(const hrp_wgrad_desc* descs, int n, void* table, hrp_batch_info* info);
int wgrad_batch_launch(const void* table_dev, const hrp_batch_info* info, hipStream_t s);
int64_t wgrad_batch_table_bytes(int n);
int wgrad_fold_prepare(const hrp_wgrad_fold_desc* descs, int n, void* table, hrp_batch_info* info);
int wgrad_fold_launch(const void* table_dev, const hrp_batch_info* info, hipStream_t s);
int64_t wgrad_fold_table_bytes(int n);

int ew_batch_prepare(int family, const void* descs, int n, void* table, hrp_batch_info* info);
int ew_batch_launch(const void* table_dev, const hrp_batch_info* info, hipStream_t s);
int64_t ew_batch_table_bytes(int family, int n);

}  // namespace hrp
