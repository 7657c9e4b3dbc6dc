// C entry points of the batched launches (include/hrp.h, hrp_batch_*): dispatch to the family's translation unit.
#include "batch.h"
#include <string.h>

using namespace hrp;

extern "C" int64_t hrp_batch_table_bytes(int family, int n) {
  if (n < 1 || n > HRP_BATCH_MAX) return 0;
  switch (family) {
    case HRP_BATCH_CONV: return conv_batch_table_bytes(n);
    case HRP_BATCH_WGRAD: return wgrad_batch_table_bytes(n);
    case HRP_BATCH_WGRAD_FOLD: return wgrad_fold_table_bytes(n);
    case HRP_BATCH_EW_FWD: case HRP_BATCH_EW_BWD_REDUCE: case HRP_BATCH_EW_BWD_APPLY: return ew_batch_table_bytes(family, n);
    default: return 0;
  }
}

extern "C" int hrp_batch_prepare(int family, const void* descs, int n, void* table_host, hrp_batch_info* info) {
  HRP_REQUIRE(descs && info, "batch: null pointer");
  HRP_REQUIRE(n >= 1 && n <= HRP_BATCH_MAX, "batch: n=%d is outside 1..%d", n, HRP_BATCH_MAX);
  memset(info, 0, sizeof(*info));
  info->family = family;
  info->n = n;
  switch (family) {
    case HRP_BATCH_CONV: {
      const hrp_conv_desc* d = (const hrp_conv_desc*)descs;
      HRP_REQUIRE(table_host, "conv batch: table_host is required");
      info->dtype = d[0].dtype;
      return d[0].dtype == HRP_F32 ? conv_batch_prepare_f32(d, n, table_host, info) :
             d[0].dtype == HRP_F32X3 ? conv_batch_prepare_f32x3(d, n, table_host, info) : conv_batch_prepare_bf16(d, n, table_host, info);
    }
    case HRP_BATCH_WGRAD:
      info->dtype = ((const hrp_wgrad_desc*)descs)[0].dtype;
      return wgrad_batch_prepare((const hrp_wgrad_desc*)descs, n, table_host, info);
    case HRP_BATCH_WGRAD_FOLD:
      HRP_REQUIRE(table_host, "wgrad fold: table_host is required");
      return wgrad_fold_prepare((const hrp_wgrad_fold_desc*)descs, n, table_host, info);
    case HRP_BATCH_EW_FWD: case HRP_BATCH_EW_BWD_REDUCE: case HRP_BATCH_EW_BWD_APPLY:
      HRP_REQUIRE(table_host, "ew batch: table_host is required");
      return ew_batch_prepare(family, descs, n, table_host, info);
    default:
      set_error("batch: unknown family %d", family);
      return HRP_ERR_ARG;
  }
}

extern "C" int hrp_batch_launch(const void* table_dev, const hrp_batch_info* info, void* stream) {
  HRP_REQUIRE(table_dev && info && info->n >= 1 && info->n <= HRP_BATCH_MAX && (info->grid > 0 || info->grid3 > 0), "batch launch: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  switch (info->family) {
    case HRP_BATCH_CONV: return info->dtype == HRP_F32 ? conv_batch_launch_f32(table_dev, info, s) :
                                 info->dtype == HRP_F32X3 ? conv_batch_launch_f32x3(table_dev, info, s) : conv_batch_launch_bf16(table_dev, info, s);
    case HRP_BATCH_WGRAD: return wgrad_batch_launch(table_dev, info, s);
    case HRP_BATCH_WGRAD_FOLD: return wgrad_fold_launch(table_dev, info, s);
    case HRP_BATCH_EW_FWD: case HRP_BATCH_EW_BWD_REDUCE: case HRP_BATCH_EW_BWD_APPLY: return ew_batch_launch(table_dev, info, s);
    default:
      set_error("batch launch: unknown family %d", info->family);
      return HRP_ERR_ARG;
  }
}
