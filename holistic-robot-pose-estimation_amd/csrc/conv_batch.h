// Batched tile convolution: n problems of one tap count and element type in one launch (hrp_batch_*, include/hrp.h).
// Every problem keeps its own tile configuration: the kernel is the union of four instances of the tile program of
// conv_tile.h behind a workgroup-uniform switch (registers / LDS of the launch = the maximum over the four).
//
// Tile choice differs from the single-problem launcher: a batch fills the chip by itself (the eight 3x3 problems of
// a stage-4 layer are 3 000+ workgroups), so every problem takes the LARGEST tile that fits - 256 pixels x 32 or 64
// output channels - instead of shrinking tiles until one problem alone reaches 256 workgroups; the weight slab of a
// 128- or 256-channel layer is then staged once per 256 pixels instead of once per 64 or 128.
#pragma once
#include "conv_row.h"
#include <string.h>

namespace hrp {

struct ConvProblem {
  hrp_conv_desc d;
  ConvTiling t;
  int cfg;        // 0: 256 px x 32 cout, 1: 128 px x 32, 2: 256 px x 64, 3: 128 px x 64, 4: 0 with persistent workgroups,
                  // 5 / 6 / 7 / 8: row-strip kernels (conv_row.h) for 32 / 64 / 128 / 256 channels; 9 / 10: whole-image kernels, 128 / 256
  int pgrid;      // cfg 4: workgroups of this problem
  int pad[2];
  RowPlan r;      // cfg 5 / 6
};

// LIGHT: the variant for batches of problems with <= 32 output channels only (the high-resolution branch of both
// trunks).  The union kernel's register allocation is the maximum over its bodies; keeping the 64-channel bodies out
// lets this one carry the persistent body (plans live across tiles: ~200 registers) and still run two workgroups
// per CU.
template <typename T, int NT, bool LIGHT>
__global__ __launch_bounds__(256, LIGHT ? 1 : 2) void conv_batch_kernel(const ConvProblem* __restrict__ tab, const BatchHdr h) {
  int base;
  const int g = batch_find(h, blockIdx.x, base);
  const ConvProblem& P = tab[g];
  const int bid = (int)blockIdx.x - base;
  const int slot = blockIdx.x & (HRP_STAT_SLOTS - 1);
  if constexpr (LIGHT) {
    switch (P.cfg) {
      case 0: conv_tile_body<T, 1, 2, 1, 4, NT, false, true>(P.d, P.t, bid, 1, slot); break;
      case 1: conv_tile_body<T, 1, 1, 1, 4, NT, false, true>(P.d, P.t, bid, 1, slot); break;
      default:   // 4: cfg 0 with P.pgrid persistent workgroups walking the problem's tiles
        if constexpr (NT == 9) conv_tile_body<T, 1, 2, 1, 4, NT, true, true>(P.d, P.t, bid, P.pgrid, slot);
        break;
    }
  } else {
    switch (P.cfg) {
      case 0: conv_tile_body<T, 1, 2, 1, 4, NT, false, true>(P.d, P.t, bid, 1, slot); break;
      case 1: conv_tile_body<T, 1, 1, 1, 4, NT, false, true>(P.d, P.t, bid, 1, slot); break;
      case 2: conv_tile_body<T, 2, 2, 1, 4, NT, false, true>(P.d, P.t, bid, 1, slot); break;
      case 5: if constexpr (NT == 9 && Elem<T>::SZ == 2) conv_row_body<32>(P.d, P.r, bid, slot); break;
      case 6: if constexpr (NT == 9 && Elem<T>::SZ == 2) conv_row_body<64>(P.d, P.r, bid, slot); break;
      case 7: if constexpr (NT == 9 && Elem<T>::SZ == 2) conv_deep_body<128>(P.d, P.r, bid, slot); break;
      case 8: if constexpr (NT == 9 && Elem<T>::SZ == 2) conv_deep_body<256>(P.d, P.r, bid, slot); break;
      case 9: if constexpr (NT == 9 && Elem<T>::SZ == 2) conv_img_body<128>(P.d, P.r, bid, slot); break;
      case 10: if constexpr (NT == 9 && Elem<T>::SZ == 2) conv_img_body<256>(P.d, P.r, bid, slot); break;
      default: conv_tile_body<T, 2, 1, 1, 4, NT, false, true>(P.d, P.t, bid, 1, slot); break;
    }
  }
}

template <typename T, int NT>
static int conv_batch_plan_one(const hrp_conv_desc& d, ConvProblem& P, int& lds) {
  int rc = -100;
  if constexpr (NT == 9 && Elem<T>::SZ == 2) {
    const int rc_ = hrp_conv_rowstrip_channels(&d);
    if (rc_) {   // the lean kernel of the high-resolution BasicBlock layers
      row_plan(d, P.r);
      P.t = ConvTiling{};
      P.t.nblocks = row_grid(P.r, rc_);
      P.cfg = rc_ == 32 ? 5 : rc_ == 64 ? 6 : rc_ == 128 ? 7 : 8;
      if (P.r.img) P.cfg += 2;
      lds = row_lds_bytes(rc_, P.r.img);
      return HRP_OK;
    }
  }
  static const int budget_kb = 76;   // (swept: DESIGN 5)
  g_conv_lds_budget_kb = budget_kb;
  if (d.Cout <= 32) {
    rc = plan_cfg<T, 1, 2, 1, 4, NT>(d, P.t, lds, false); P.cfg = 0;
    if (rc == -100) { rc = plan_cfg<T, 1, 1, 1, 4, NT>(d, P.t, lds, false); P.cfg = 1; }
  } else {
    rc = plan_cfg<T, 2, 2, 1, 4, NT>(d, P.t, lds, false); P.cfg = 2;
    if (rc == -100) { rc = plan_cfg<T, 2, 1, 1, 4, NT>(d, P.t, lds, false); P.cfg = 3; }
    if (rc == -100) { rc = plan_cfg<T, 1, 2, 1, 4, NT>(d, P.t, lds, false); P.cfg = 0; }
    if (rc == -100) { rc = plan_cfg<T, 1, 1, 1, 4, NT>(d, P.t, lds, false); P.cfg = 1; }
  }
  if (rc == -100) {
    set_error("conv batch: no tile configuration fits LDS and the tap halo (H=%d W=%d Cin=%d stride=%d taps=%d)", d.H, d.W, d.Cin, d.in_stride, d.ntaps);
    return HRP_ERR_ARG;
  }
  return rc;
}

template <typename T, int NT>
static int conv_batch_prepare_nt(const hrp_conv_desc* descs, int n, ConvProblem* tab, hrp_batch_info* info) {
  constexpr int SZ = Elem<T>::SZ;
  ConvProblem probs[HRP_BATCH_MAX];
  long weight[HRP_BATCH_MAX];
  int lds_max = 0;
  for (int i = 0; i < n; ++i) {
    const hrp_conv_desc& d = descs[i];
    HRP_REQUIRE((d.Cin * SZ) % ROW == 0, "conv batch: Cin * sizeof(T) must be a multiple of 32 bytes (Cin=%d)", d.Cin);
    memset(&probs[i], 0, sizeof(ConvProblem));
    probs[i].d = d;
    int lds = 0;
    const int rc = conv_batch_plan_one<T, NT>(d, probs[i], lds);
    if (rc != HRP_OK) return rc;
    lds_max = lds > lds_max ? lds : lds_max;
    // work of one workgroup (MFMA steps): the long-running problems go first so that the launch tail is short
    const int ct = probs[i].cfg >= 2 ? 2 : 1, pt = (probs[i].cfg & 1) ? 1 : 2;
    weight[i] = (long)cdiv(d.Cin * SZ, ROW) * NT * ct * pt;
    if (probs[i].cfg >= 5) weight[i] = (long)cdiv(d.Cin * SZ, ROW) * NT * (probs[i].cfg == 9 ? 8 : 4);     // tiles per wave
  }
  int order[HRP_BATCH_MAX];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i)   // stable insertion sort, heaviest first
    for (int j = i; j > 0 && weight[order[j]] > weight[order[j - 1]]; --j) { int t_ = order[j]; order[j] = order[j - 1]; order[j - 1] = t_; }
  // A batch of small-channel problems only runs the LIGHT kernel.  Its 3x3 problems with many tiles of two K chunks
  // (the index arithmetic of such a tile costs more instructions than its MFMAs) get persistent workgroups that keep
  // their DMA / store plans across tiles: two per CU over the launch.
  bool light = true;
  for (int i = 0; i < n; ++i) light = light && probs[i].cfg <= 1;
  static const int pwgs = 512;
  const int pcap = light && pwgs > 0 ? (pwgs / n < 32 ? 32 : pwgs / n) : 0;
  int blk = 0;
  for (int k = 0; k < n; ++k) {
    ConvProblem& P = probs[order[k]];
    info->blk0[k] = blk;
    if (NT == 9 && pcap > 0 && P.cfg == 0 && P.t.nblocks >= 2 * pcap && P.d.Cin * SZ <= 2 * ROW && !P.d.bnb_x) {
      P.cfg = 4;
      P.pgrid = pcap;
      blk += pcap;
    } else {
      blk += P.t.nblocks;
    }
    if (tab) tab[k] = P;
  }
  info->blk0[n] = blk;
  info->grid = blk;
  info->lds_bytes = lds_max;
  info->variant = NT + (light ? 100 : 0);
  return HRP_OK;
}

template <typename T>
static int conv_batch_prepare_t(const hrp_conv_desc* descs, int n, void* table, hrp_batch_info* info) {
  for (int i = 0; i < n; ++i) {
    const int rc = conv_check(&descs[i]);
    if (rc != HRP_OK) return rc;
    HRP_REQUIRE(descs[i].ntaps == descs[0].ntaps && descs[i].dtype == descs[0].dtype, "conv batch: mixed tap counts / element types");
    // a pointwise problem runs the tile program inside a batch: only the forms that program knows
    HRP_REQUIRE(descs[i].ntaps != 1 || !descs[i].bnb_x || (descs[i].bnb_mask && descs[i].bnb_consts && !descs[i].res),
                "conv batch: a 1x1 problem with a mask-less / residual epilogue reduce must be launched on its own (hrp_conv_pointwise)");
    // skinny fp32 linear layers take the split-K path of the single launcher (memset + atomics): not batched
    HRP_REQUIRE(!(descs[i].dtype == HRP_F32 && descs[i].H == 1 && descs[i].W == 1 && descs[i].Cin >= 512),
                "conv batch: linear layers are launched one by one");
  }
  ConvProblem* tab = (ConvProblem*)table;
  switch (descs[0].ntaps) {
    case 1: return conv_batch_prepare_nt<T, 1>(descs, n, tab, info);
    case 2: return conv_batch_prepare_nt<T, 2>(descs, n, tab, info);
    case 4: return conv_batch_prepare_nt<T, 4>(descs, n, tab, info);
    case 9: return conv_batch_prepare_nt<T, 9>(descs, n, tab, info);
    default:
      set_error("conv batch: ntaps=%d is not one of the batched tap counts (1, 2, 4, 9)", descs[0].ntaps);
      return HRP_ERR_ARG;
  }
}

template <typename T, int NT, bool LIGHT>
static int conv_batch_launch_nt(const void* table_dev, const hrp_batch_info* info, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_batch_kernel<T, NT, LIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const BatchHdr h = make_hdr(info->blk0, info->n);
  hipLaunchKernelGGL((conv_batch_kernel<T, NT, LIGHT>), dim3(info->grid), dim3(256), info->lds_bytes, s, (const ConvProblem*)table_dev, h);
  return check_launch("conv_batch_kernel");
}

template <typename T>
static int conv_batch_launch_t(const void* table_dev, const hrp_batch_info* info, hipStream_t s) {
  switch (info->variant) {
    case 1: return conv_batch_launch_nt<T, 1, false>(table_dev, info, s);
    case 2: return conv_batch_launch_nt<T, 2, false>(table_dev, info, s);
    case 4: return conv_batch_launch_nt<T, 4, false>(table_dev, info, s);
    case 9: return conv_batch_launch_nt<T, 9, false>(table_dev, info, s);
    case 101: return conv_batch_launch_nt<T, 1, true>(table_dev, info, s);
    case 102: return conv_batch_launch_nt<T, 2, true>(table_dev, info, s);
    case 104: return conv_batch_launch_nt<T, 4, true>(table_dev, info, s);
    case 109: return conv_batch_launch_nt<T, 9, true>(table_dev, info, s);
    default: set_error("conv batch: bad variant %d", info->variant); return HRP_ERR_ARG;
  }
}

}  // namespace hrp
