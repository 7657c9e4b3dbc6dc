// Host side of the fused row-strip backward launches (include/hrp.h, hrp_rowbw_*); kernels: conv_rowbw.h, instantiated per form
// in conv_rowbw_f0.hip .. conv_rowbw_f3.hip.
#include "conv_rowbw.h"

namespace hrp {

extern template int rowbw_launch_form<0>(const RowBwArgs&, int, int, hipStream_t);
extern template int rowbw_launch_form<1>(const RowBwArgs&, int, int, hipStream_t);
extern template int rowbw_launch_form<2>(const RowBwArgs&, int, int, hipStream_t);
extern template int rowbw_launch_form<3>(const RowBwArgs&, int, int, hipStream_t);

#ifdef HRP_TIMELINE
extern template int rowbw_timeline_form<0>(void*, int, int);
extern template int rowbw_timeline_form<1>(void*, int, int);
extern template int rowbw_timeline_form<2>(void*, int, int);
extern template int rowbw_timeline_form<3>(void*, int, int);
#endif

// which kernel form runs this problem (conv_rowbw.h, rowbw_body): the specialised forms carry no run-time options
static int rowbw_form(const hrp_rowbw_desc& q) {
  const hrp_conv_desc& d = q.conv;
  static const bool generic_only = getenv("HRP_ROWBW_GENERIC") != nullptr;      // A/B switch
  if (generic_only || d.pro_mode != 2 || d.pro_side || d.pro_side2 || d.scale || d.relu) return 0;
  if (d.pro_mask && d.bnb_x && !d.bnb_mask && !d.res && d.stats) return 1;
  if (!d.pro_mask && d.res && !d.bnb_x && !d.stats) return 2;
  if (!d.pro_mask && d.res && d.bnb_x && d.bnb_mask && d.stats) return 3;
  return 0;
}

static int rowbw_channels(const hrp_rowbw_desc& q) {
  const int C = row_channels(q.conv);
  if (C != 32 && C != 64) return 0;
  if (!q.wg_x || !q.dw || (uintptr_t)q.wg_x % 16 || (uintptr_t)q.dw % 16) return 0;
  if (q.wg_act && (!q.conv.bnb_stats || !q.conv.bnb_gamma || !q.conv.bnb_beta)) return 0;
  if ((uintptr_t)q.conv.pro_mask % 4) return 0;      // (mask bytes are fetched as aligned dwords)
  return C;
}

}  // namespace hrp

using namespace hrp;

extern "C" int hrp_rowbw_channels(const hrp_rowbw_desc* d) {
  static const bool off = getenv("HRP_NO_ROWBW") != nullptr;
  return (d && !off) ? rowbw_channels(*d) : 0;
}

extern "C" int64_t hrp_rowbw_table_bytes(void) { return (int64_t)sizeof(RowBwArgs); }

extern "C" int hrp_rowbw_prepare(const hrp_rowbw_desc* descs, int n, int max_wgs, void* table, hrp_rowbw_info* info) {
  HRP_REQUIRE(descs && info, "rowbw: null pointer");
  HRP_REQUIRE(n >= 1 && n <= HRP_ROWBW_MAX, "rowbw: n=%d is outside 1..%d", n, HRP_ROWBW_MAX);
  memset(info, 0, sizeof(*info));
  int total = 0, lds = 0, nstr = 0;
  HRP_REQUIRE(n == 1 || (rowbw_channels(descs[0]) == 32 && rowbw_channels(descs[1]) == 64 && rowbw_form(descs[0]) == rowbw_form(descs[1])),
              "rowbw: two problems of one launch are a 32-channel and a 64-channel one of the same form, in this order");
  // cost of a strip in the split of the launch (conv_rowbw.h, RowBwArgs): measured 8.9 us (C = 32) against 12.2 us (C = 64)
  static const int cost64 = getenv("HRP_ROWBW_COST64") ? atoi(getenv("HRP_ROWBW_COST64")) : 14;
  int cost0[HRP_ROWBW_MAX + 1], cost[HRP_ROWBW_MAX], ns[HRP_ROWBW_MAX];
  for (int i = 0; i < n; ++i) {
    const int C = rowbw_channels(descs[i]);
    HRP_REQUIRE(C != 0, "rowbw: problem %d is not a 32- / 64-channel row-strip data gradient with wg_x / dw set", i);
    const int rc = conv_check(&descs[i].conv);
    if (rc != HRP_OK) return rc;
    info->strip0[i] = nstr;
    ns[i] = descs[i].conv.N * (descs[i].conv.H / 8);
    cost[i] = C == 32 ? 10 : (cost64 > 0 ? cost64 : 14);
    cost0[i] = total;
    total += ns[i] * cost[i];
    nstr += ns[i];
    const int l = C == 32 ? BwCfg<32>::LDS_BYTES : BwCfg<64>::LDS_BYTES;
    lds = l > lds ? l : lds;
  }
  cost0[n] = total;
  info->strip0[n] = nstr;
  if (max_wgs <= 0) {
    static const int env = getenv("HRP_ROWBW_WGS") ? atoi(getenv("HRP_ROWBW_WGS")) : 0;
    max_wgs = env;
    if (max_wgs <= 0) {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        max_wgs = cus;
      else
        max_wgs = 256;
      (void)hipGetLastError();
    }
  }
  // (every workgroup interval is at least one strip of the costliest problem long: no workgroup between a problem's first and
  // last one is left without a strip - its slab would stay unwritten)
  int maxc = 0;
  for (int i = 0; i < n; ++i) maxc = cost[i] > maxc ? cost[i] : maxc;
  int nwg = nstr < max_wgs ? nstr : max_wgs;
  if (nwg > total / maxc) nwg = total / maxc;
  info->n = n; info->grid = nwg; info->lds_bytes = lds; info->total_strips = nstr;
  for (int i = 0; i < n; ++i) {
    int first = -1, last = -1;
    for (int w = 0; w < nwg; ++w) {
      const int lo = (int)((long long)w * total / nwg), hi = (int)((long long)(w + 1) * total / nwg);
      int k0, k1;
      rowbw_range(lo, hi, cost0[i], cost[i], ns[i], k0, k1);
      if (k0 < k1) { if (first < 0) first = w; last = w; }
    }
    HRP_REQUIRE(first >= 0, "rowbw: problem %d got no workgroup", i);
    info->first_wg[i] = first;
    info->G[i] = last - first + 1;
    const int C = rowbw_channels(descs[i]);
    info->ws_bytes[i] = (int64_t)info->G[i] * (C / 32) * (C / 32) * 9 * 1024 * 4;
  }
  if (table) {
    RowBwArgs& A = *(RowBwArgs*)table;
    memset(&A, 0, sizeof(A));
    for (int i = 0; i < n; ++i) {
      HRP_REQUIRE(descs[i].workspace && ((uintptr_t)descs[i].workspace % 16) == 0 && descs[i].workspace_bytes >= info->ws_bytes[i],
                  "rowbw: problem %d needs %lld bytes of 16-byte aligned workspace", i, (long long)info->ws_bytes[i]);
      A.q[i] = descs[i];
      row_plan(descs[i].conv, A.rp[i]);
      A.first_wg[i] = info->first_wg[i];
      A.cost0[i] = cost0[i]; A.cost[i] = cost[i]; A.nstrips[i] = ns[i];
    }
    A.cost0[n] = total;
    A.n = n; A.total = total; A.nwg = nwg;
  }
  return HRP_OK;
}

extern "C" int hrp_rowbw_fold_descs(const hrp_rowbw_desc* descs, const hrp_rowbw_info* info, hrp_wgrad_fold_desc* out) {
  HRP_REQUIRE(descs && info && out && info->n >= 1 && info->n <= HRP_ROWBW_MAX, "rowbw fold: bad arguments");
  for (int i = 0; i < info->n; ++i) {
    const int C = rowbw_channels(descs[i]);
    HRP_REQUIRE(C != 0 && descs[i].workspace && descs[i].workspace_bytes >= info->ws_bytes[i], "rowbw fold: problem %d has no workspace", i);
    hrp_wgrad_fold_desc f;
    memset(&f, 0, sizeof(f));
    f.workspace = (const float*)descs[i].workspace; f.dw = descs[i].dw;
    f.G = info->G[i]; f.pairs = (C / 32) * (C / 32); f.n_cib = C / 32; f.nte = 9; f.nb = 1;
    f.Cout = C; f.dw_cin = C; f.ntaps = 9; f.dw_tap_stride = 0; f.dw_tap_off = 0; f.accumulate = descs[i].accumulate;
    out[i] = f;
  }
  return HRP_OK;
}

extern "C" int hrp_rowbw_launch(const void* table, const hrp_rowbw_info* info, void* stream) {
  HRP_REQUIRE(table && info && info->n >= 1 && info->n <= HRP_ROWBW_MAX && info->grid > 0, "rowbw launch: bad arguments");
  const RowBwArgs& A = *(const RowBwArgs*)table;
  HRP_REQUIRE(A.n == info->n && A.nwg == info->grid, "rowbw launch: table and info do not belong together");
  hipStream_t s = (hipStream_t)stream;
  switch (rowbw_form(A.q[0])) {
    case 1: return rowbw_launch_form<1>(A, info->grid, info->lds_bytes, s);
    case 2: return rowbw_launch_form<2>(A, info->grid, info->lds_bytes, s);
    case 3: return rowbw_launch_form<3>(A, info->grid, info->lds_bytes, s);
    default: return rowbw_launch_form<0>(A, info->grid, info->lds_bytes, s);
  }
}

#ifdef HRP_TIMELINE
extern "C" int hrp_debug_rowbw_timeline(void* dst, int nblocks, int clear, int form) {
  switch (form) {
    case 1: return rowbw_timeline_form<1>(dst, nblocks, clear);
    case 2: return rowbw_timeline_form<2>(dst, nblocks, clear);
    case 3: return rowbw_timeline_form<3>(dst, nblocks, clear);
    default: return rowbw_timeline_form<0>(dst, nblocks, clear);
  }
}
extern "C" int hrp_debug_rowbw_form(const hrp_rowbw_desc* d) { return hrp::rowbw_form(*d); }
#endif
