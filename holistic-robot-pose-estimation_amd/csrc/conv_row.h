// Row-strip 3x3 convolution for the high-resolution branches of HRNet (gfx950, bf16): the lean kernel of the
// BasicBlock layers conv3x3 C -> C with C = 32 @ W = 64 and C = 64 @ W = 32 (reference HRnet.py:28-57; 130 of the 326
// convolutions of an HRNet-W32 forward and as many data gradients).  In both shapes an image row is exactly 4 KiB.
//
// Why a second kernel: the general tile program (conv_tile.h) spends ~2 100 instructions per wave on a 256-pixel tile of
// these layers (index arithmetic of its DMA / store plans, LDS round trip of the epilogue) for 36 MFMAs; here the shape
// is a template constant and everything is affine in the lane id:
//   workgroup (4 waves) -> one strip of TH = 8 full-width output rows of one image, all output channels
//   staging   : the TH + 2 input rows are CONTIGUOUS in NHWC memory: 10 x 4 direct-to-LDS DMA pieces of 1 KiB, source =
//               row base + lane constant (no bounds logic: rows outside the image are zero-filled, the left / right
//               padding is ONE zero pixel between consecutive LDS rows, shared by x = W of row r and x = -1 of row r + 1)
//   weights   : the wave's 32 output channels x all taps x all input channels live in REGISTERS as MFMA A fragments
//               (72 VGPRs for C = 32, 144 for C = 64), read once per workgroup from the packed layout of hrp_pack_weights
//   MFMA loop : a wave owns 4 vertically adjacent 32-pixel tiles; every B fragment (one ds_read_b128 of 32 pixels x 16
//               channels at a tap-shifted address) feeds up to three accumulators (the taps dy = -1, 0, +1 of the three
//               output rows that see this input row): 0.5 LDS reads per MFMA, no weight reads in the loop
//   epilogue  : the MFMA row -> output channel assignment is permuted (a free choice: it is just which weight row a lane
//               loads) so that a lane's 16 accumulators are 16 CONSECUTIVE channels of one pixel: two 16-byte stores per
//               tile straight from registers, no LDS round trip; per-channel statistics are reduced across lanes with a
//               reduce-scatter butterfly (31 shuffles for 32 values) once per workgroup
//   LDS bank conflicts: a 16-lane group of ds_read_b128 reads the same 16-byte slot of 16 different pixels (pixel stride
//               64 / 128 bytes = 4 / 2 pixels per 256-byte bank row); slot' = slot ^ g(x) with g = (x >> 2) & 3 resp.
//               (x >> 1) & 7 of the IMAGE column x spreads them over all 16 slots.  The DMA writes lane-linear, so the
//               permutation is applied on the per-lane source address (a constant) and again on the read address.
//
// Fused BatchNorm work (train mode; what takes the element-wise passes of a BasicBlock interior off HBM):
//   pro_mode 1   the staged rows are x' = relu(bn(x)) (statistics of x from its producer's epilogue): every lane
//                transforms, in place, exactly the 16 bytes it DMA'd itself (no extra barrier), and also stores them to
//                pro_side (the activation the weight gradient of this layer reads) - replaces hrp_ew_fwd
//   pro_mode 2   the staged rows are the BatchNorm + ReLU BACKWARD of (x = gradient of the activation, pro_x2 = BatchNorm
//                input, same lane-constant addressing, through registers) - replaces hrp_ew_bwd_apply
//   bnb_x        epilogue: sum g, sum g * xhat of the stored gradient (mask recomputed from bnb_x) - replaces
//                hrp_ew_bwd_reduce (as conv_tile.h's bnb_*, but without the separate mask tensor)
#pragma once
#include "conv_tile.h"
#include <type_traits>

namespace hrp {

template <int C>
struct RowCfg {
  static constexpr int W = 2048 / C;        // 64, 32
  static constexpr int P = 2 * C;           // bytes per pixel
  static constexpr int S = P / 16;          // 16-byte slots per pixel
  static constexpr int KS = C / 16;         // MFMA k-steps per tap
  static constexpr int MT = C / 32;         // 32-channel output tiles
  static constexpr int NCOL = W / 32;       // 32-pixel tiles per image row
  static constexpr int TH = 8;              // output rows per workgroup
  static constexpr int PXP = 1024 / P;      // pixels per 1 KiB DMA piece
  static constexpr int ROWB = (W + 1) * P;  // LDS row pitch: the row + one zero pixel
  static constexpr int NROWS = TH + 2;
  static constexpr int TILE_BYTES = P + NROWS * ROWB;               // leading zero pixel + rows
  static constexpr int CTAB_OFF = (TILE_BYTES + 255) & ~255;        // per-channel constants [10][C] floats
  static constexpr int STAT_OFF = CTAB_OFF + 10 * C * 4;             // [4 waves][64] floats
  static constexpr bool PERSIST = C == 32;                          // persistent workgroups, weights parked in LDS (18 KiB)
  static constexpr int WLDS_OFF = STAT_OFF + 4 * 64 * 4;            // [9 taps][KS] A fragments of 1 KiB, lane linear
  static constexpr int LDS_BYTES = WLDS_OFF + (PERSIST ? 9 * KS * 1024 : 0);
  static_assert(MT * NCOL == 2, "two waves side by side (columns or channel tiles), two on top of each other");
  __device__ static __forceinline__ int g(int x) { return C == 32 ? (x >> 2) & 3 : (x >> 1) & 7; }
};

struct RowPlan {
  int wslot[9];        // packed-weight tap slot of the canonical tap (dy + 1) * 3 + (dx + 1)
  int nstrips, spi;    // workgroups (strip kernels: N * H / TH); strips per image
  FastDiv fd_spi;
  int img;             // 1: whole-image variant (conv_img_body): C = 128 @ 16 x 16, C = 256 @ 8 x 8
  int spw;             // conv_row_body: strips per (persistent) workgroup
};

// mean / invstd / scale / shift of one channel from the statistic slots (the arithmetic of elementwise.hip's
// channel_consts: m = sum / n, var = max(sumsq / n - m^2, 0), invstd = rsqrt(var + eps), sc = gamma * invstd, sh = beta - m * sc)
__device__ __forceinline__ void row_bn_consts(const double* stats, const float* gamma, const float* beta, float count, float eps,
                                              int c, int C, float& mean, float& inv, float& sc, float& sh) {
  const float m = slot_sum(stats, c, 2 * C) / count;
  const float var = fmaxf(slot_sum(stats, C + c, 2 * C) / count - m * m, 0.f);
  inv = rsqrtf(var + eps);
  mean = m;
  sc = gamma[c] * inv;
  sh = beta[c] - m * sc;
}

// the relu(bn(.)) of the forward prologue and the mask of both backward uses: ONE expression, so that the three agree bit for bit
__device__ __forceinline__ float row_bn_act(float x, float sc, float sh) { return fmaf(x, sc, sh); }

// Host: is this problem one the row-strip kernel takes?  (Everything else runs the general tile program.)
static inline int row_channels(const hrp_conv_desc& d) {
  if (d.dtype != HRP_BF16 || d.ntaps != 9 || d.in_stride != 1 || d.out_stride != 1) return 0;
  if (d.Cin != d.Cout || (d.Cin != 32 && d.Cin != 64 && d.Cin != 128 && d.Cin != 256)) return 0;
  const int C = d.Cin;
  if (d.W != 2048 / C || d.Wo != d.W || d.Ho != d.H || d.H % 8 || d.y_H != d.Ho || d.y_W != d.Wo) return 0;
  if (d.x_pitch != C || d.y_pitch != C || d.w_cout_pad != C || (d.res && d.res_pitch != C)) return 0;
  if (((uintptr_t)d.x | (uintptr_t)d.y | (uintptr_t)d.w | (uintptr_t)d.res) % 16) return 0;
  if (d.bias) return 0;
  if (d.bnb_x && (d.bnb_x_pitch != C || !d.bnb_stats || !d.bnb_gamma || !d.bnb_beta || !d.stats ||
                  d.relu || d.scale || (uintptr_t)d.bnb_x % 16)) return 0;
  if (d.bnb_x && d.bnb_mask && (d.bnb_mask_pitch != C / 8 || (uintptr_t)d.bnb_mask % 2)) return 0;
  if (d.pro_mode < 0 || d.pro_mode > 3) return 0;
  if (d.pro_mode && (!d.pro_stats || !d.pro_gamma || !d.pro_beta)) return 0;
  if (d.pro_mode == 2 && (!d.pro_x2 || !d.pro_bsums || (uintptr_t)d.pro_x2 % 16)) return 0;
  // pro_mode 3 (x = raw conv2 output of the previous block, pro_x2 = that block's input): the activation goes to pro_side, its
  // ReLU bits to pro_mask (an OUTPUT here) - both required, the backward of the previous block reads them
  if (d.pro_mode == 3 && (!d.pro_x2 || (uintptr_t)d.pro_x2 % 16 || !d.pro_side || !d.pro_mask || d.pro_side2)) return 0;
  if (d.pro_mode < 2 && (d.pro_mask || d.pro_side2)) return 0;
  if ((d.pro_side && (uintptr_t)d.pro_side % 16) || (d.pro_side2 && (uintptr_t)d.pro_side2 % 16)) return 0;
  if (d.res_mask && (!d.res || (uintptr_t)d.res_mask % 2 || d.relu || d.scale)) return 0;
  if ((long long)d.N * d.H * d.W * C * 2 >= (1ll << 31)) return 0;     // 32-bit byte offsets inside the tensors
  unsigned seen = 0;
  for (int i = 0; i < 9; ++i) {
    if (d.dy[i] < -1 || d.dy[i] > 1 || d.dx[i] < -1 || d.dx[i] > 1) return 0;
    seen |= 1u << ((d.dy[i] + 1) * 3 + d.dx[i] + 1);
  }
  return seen == 0x1ff ? C : 0;
}

static inline void row_plan(const hrp_conv_desc& d, RowPlan& rp) {
  for (int i = 0; i < 9; ++i) rp.wslot[(d.dy[i] + 1) * 3 + d.dx[i] + 1] = d.wtap[i];
  rp.spi = d.H / 8;
  rp.nstrips = d.N * rp.spi;
  rp.fd_spi = make_fastdiv(rp.spi);
  rp.img = 0;
  // persistent workgroups of the 32 / 64-channel kernel: 2 strips each once the problem has more strips than the chip
  // has workgroup slots for it (B = 64: 512 / 256 strips -> 256 / 128 workgroups)
  // (one strip per workgroup at B = 64: conv family -0.6 ms one by one, nothing in the step - DESIGN 5, round 5)
  rp.spw = rp.nstrips >= 256 ? 2 : 1;
  if (rp.spi % rp.spw || d.Cin != 32) rp.spw = 1;       // (the 64-channel kernel keeps its weights in registers: one strip)
  static const bool no_img = false;
  if (!no_img && (d.Cin == 128 || d.Cin == 256) && d.H == d.W) {
    rp.img = 1;
    rp.nstrips = d.Cin == 128 ? d.N : ((d.N + 1) / 2) * 2;
  }
}


// ---- shared prologue pieces: the per-channel constants of a lane's 8 channels (one 16-byte slot) and the two transforms ----
// v & (bit i of bits ? ~0 : 0) on the bit pattern of a float: one signed bit-field extract + one AND (a compare + select pair
// costs three instructions and a VCC round trip)
__device__ __forceinline__ float row_keep_if_bit(const float v, const int bits, const int i) {
  const int m = (int)((unsigned)bits << (31 - i)) >> 31;
  return __int_as_float(__float_as_int(v) & m);
}

struct RowPro {
  float sc[8], sh[8], c1[8], c0[8];
  __device__ __forceinline__ void load(const float* ctab, const int C, const int cb) {
#pragma unroll
    for (int i = 0; i < 8; i += 4) {
      const float4 va = *(const float4*)(ctab + cb + i), vb = *(const float4*)(ctab + C + cb + i);
      sc[i] = va.x; sc[i + 1] = va.y; sc[i + 2] = va.z; sc[i + 3] = va.w;
      sh[i] = vb.x; sh[i + 1] = vb.y; sh[i + 2] = vb.z; sh[i + 3] = vb.w;
    }
  }
  // ... and of pro_mode 2 (called inside that branch only: as a conditional part of load() the extra values cost the 32-channel
  // kernel 25 spilled registers).  Table rows 2 .. 5: a = invstd, b = -mean * invstd, k0 = sum g / n, k1 = sum g xhat / n.  The
  // BatchNorm backward  sc * (g - k0 - xhat * k1),  xhat = a x + b,  is evaluated as  sc * g + (c1 * x + c0)  with
  // c1 = -sc k1 a,  c0 = -sc (k0 + k1 b): two FMAs per element.
  __device__ __forceinline__ void load2(const float* ctab, const int C, const int cb) {
#pragma unroll
    for (int i = 0; i < 8; i += 4) {
      const float4 va = *(const float4*)(ctab + 2 * C + cb + i), vb = *(const float4*)(ctab + 3 * C + cb + i);
      const float4 v0 = *(const float4*)(ctab + 4 * C + cb + i), v1 = *(const float4*)(ctab + 5 * C + cb + i);
      const float a[4] = {va.x, va.y, va.z, va.w}, b[4] = {vb.x, vb.y, vb.z, vb.w};
      const float k0[4] = {v0.x, v0.y, v0.z, v0.w}, k1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c1[i + e] = -sc[i + e] * k1[e] * a[e];
        c0[i + e] = -sc[i + e] * fmaf(k1[e], b[e], k0[e]);
      }
    }
  }
  // pro_mode 1: relu(bn(x))
  __device__ __forceinline__ uint4 act(const uint4 raw) const {
    float f[8];
    Elem<bf16_t>::unpack(raw, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = fmaxf(row_bn_act(f[i], sc[i], sh[i]), 0.f);
    return Elem<bf16_t>::pack(f);
  }
  // pro_mode 3: the block-end activation of the PREVIOUS BasicBlock, relu(bn(y2) + res) (HRnet.py:52-56), applied while this block's
  // conv1 stages its input; bits: bit i = channel i of the vector is > 0 (the layout hrp_ew_fwd writes)
  __device__ __forceinline__ uint4 fwd3(const uint4 raw, const uint4 res, unsigned& bits) const {
    float f[8], r[8];
    Elem<bf16_t>::unpack(raw, f);
    Elem<bf16_t>::unpack(res, r);
    bits = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f[i] = fmaxf(row_bn_act(f[i], sc[i], sh[i]) + r[i], 0.f);
      bits |= (f[i] > 0.f ? 1u : 0u) << i;
    }
    return Elem<bf16_t>::pack(f);
  }
  // pro_mode 2: BatchNorm + ReLU backward of (gradient of the activation, BatchNorm input).  The ReLU mask: recomputed from
  // the BatchNorm input or (BITS) the bit mask hrp_ew_fwd wrote (bit i = channel i of the vector was > 0).
  // gm (GM): the masked gradient g itself (what an identity / residual input of the same activation receives).
  template <bool BITS, bool GM>
  __device__ __forceinline__ uint4 bwd_t(const uint4 graw, const uint4 xraw, const int bits, uint4& gm) const {
    float gq[8], xv[8];
    Elem<bf16_t>::unpack(graw, gq);
    Elem<bf16_t>::unpack(xraw, xv);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (BITS) gq[i] = row_keep_if_bit(gq[i], bits, i);
      else gq[i] = row_bn_act(xv[i], sc[i], sh[i]) > 0.f ? gq[i] : 0.f;
    }
    if constexpr (GM) gm = Elem<bf16_t>::pack(gq);
#pragma unroll
    for (int i = 0; i < 8; ++i) gq[i] = fmaf(sc[i], gq[i], fmaf(c1[i], xv[i], c0[i]));
    return Elem<bf16_t>::pack(gq);
  }
  // ub / wgm: WORKGROUP-UNIFORM choices (descriptor pointers): the mask comes as bits; the masked gradient is wanted.  They
  // select a code version per call with a scalar branch (as a per-element `bits < 0 ? .. : ..` the choice compiled to an
  // exec-mask branch pair per element: 1 200 scalar instructions per strip).
  template <bool EXT>
  __device__ __forceinline__ uint4 bwd(const uint4 graw, const uint4 xraw, const int bits, uint4& gm, const bool ub, const bool wgm) const {
    if constexpr (!EXT) return bwd_t<false, false>(graw, xraw, bits, gm);
    else {
      if (ub) return wgm ? bwd_t<true, true>(graw, xraw, bits, gm) : bwd_t<true, false>(graw, xraw, bits, gm);
      return wgm ? bwd_t<false, true>(graw, xraw, bits, gm) : bwd_t<false, false>(graw, xraw, bits, gm);
    }
  }
};

// second side output of pro_mode 2: the masked gradient, written or accumulated
__device__ __forceinline__ void row_side2(const hrp_conv_desc& d, const unsigned off, const uint4 gm) {
  char* q = (char*)d.pro_side2 + off;
  if (d.pro_side2_acc) {
    float o[8], g[8];
    Elem<bf16_t>::unpack(*(const uint4*)q, o);
    Elem<bf16_t>::unpack(gm, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] += g[i];
    *(uint4*)q = Elem<bf16_t>::pack(o);
  } else {
    *(uint4*)q = gm;
  }
}

// ---- shared epilogue pieces ------------------------------------------------------------------------------------------
// The lane holds, for each of NT tiles, 16 consecutive output channels (cl .. cl + 15) of one pixel; off[t] = byte offset of
// those 32 bytes inside y (the same offset addresses res and bnb_x: same geometry).  okmask bit t clear: tile t lies outside
// the tensor (its accumulators are zero: nothing is stored, nothing is read).  Options of hrp_conv_desc: folded-BatchNorm
// affine, residual (res == y: accumulate), ReLU; statistics of the values as stored - sum / sum of squares, or (bnb) the
// BatchNorm-backward sums of the stored gradient g masked by [bn(bnb_x) > 0]: s1 += g, s2 += g * bnb_x (finished to
// sum g * xhat = a * s2 + b * s1 by the caller).  ctab rows 8 / 9 hold the mask's scale / shift (bnb).
// The extended options - ReLU masks given as bits (pro_mask, bnb_mask), the second side output, a residual under the epilogue
// reduce - live in a second instantiation of every body (EXT): the lean one keeps the registers and schedule of the common
// launches (measured: the options as run-time checks cost the plain block-interior launches 5 - 12 %).
__device__ __forceinline__ bool row_ext(const hrp_conv_desc& d) {
  return d.pro_mask != nullptr || d.pro_side2 != nullptr || d.res_mask != nullptr ||
         (d.bnb_x != nullptr && (d.res != nullptr || d.bnb_mask != nullptr));
}

// The epilogue reduce (bnb): y = gradient of act = relu(bn(bnb_x)) (+ res: the last of several producers accumulates onto the
// others'), stored unmasked; sums of the stored value under the mask - recomputed from bnb_x (interior of a block) or (BITS) the bit
// mask hrp_ew_fwd wrote (block outputs).  RES: a residual (optionally under its own bit mask, res_mask) is added first.  A tile
// outside the tensor (okmask bit clear) has zero accumulators and loads nothing: its masked values are zero without a test.
// Operands of the epilogue forms that read tensors (the reduce's BatchNorm input and bit mask, a residual and its bit mask):
// loaded by row_epi_load, consumed by row_epi_bnb_math.
template <int NT>
struct RowEpiOps {
  uint4 xr[NT][2], rr[NT][2];
  int mb[NT], rb[NT];
};

template <int NT, bool BNB, bool BITS, bool RES>
__device__ __forceinline__ void row_epi_load(const hrp_conv_desc& d, const unsigned (&off)[NT], const unsigned okmask, RowEpiOps<NT>& e) {
  const char* bx = (const char*)d.bnb_x;
  const char* rq = (const char*)d.res;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    e.xr[t][0] = e.xr[t][1] = e.rr[t][0] = e.rr[t][1] = make_uint4(0, 0, 0, 0);
    e.mb[t] = 0;
    e.rb[t] = 0xffff;
    if ((okmask >> t) & 1) {
      if constexpr (BNB) { e.xr[t][0] = *(const uint4*)(bx + off[t]); e.xr[t][1] = *(const uint4*)(bx + off[t] + 16); }
      if constexpr (BNB && BITS) e.mb[t] = *(const unsigned short*)(d.bnb_mask + (off[t] >> 4));
      if constexpr (RES) {
        e.rr[t][0] = *(const uint4*)(rq + off[t]); e.rr[t][1] = *(const uint4*)(rq + off[t] + 16);
        if (d.res_mask) e.rb[t] = *(const unsigned short*)(d.res_mask + (off[t] >> 4));
      }
    }
  }
}

// The epilogue reduce (bnb): y = gradient of act = relu(bn(bnb_x)) (+ res: the last of several producers accumulates onto the
// others'), stored unmasked; sums of the stored value under the mask - recomputed from bnb_x (interior of a block) or (BITS) the bit
// mask hrp_ew_fwd wrote (block outputs).  RES: a residual (optionally under its own bit mask, res_mask) is added first.  A tile
// outside the tensor (okmask bit clear) has zero accumulators and loads nothing: its masked values are zero without a test.
template <int NT, bool BITS, bool RES>
__device__ __forceinline__ void row_epi_bnb_math(const hrp_conv_desc& d, const f32x16 (&acc)[NT], const unsigned (&off)[NT],
                                                 const unsigned okmask, const int cl, const float* ctab, const int C,
                                                 const RowEpiOps<NT>& e, float (&s1)[16], float (&s2)[16]) {
  char* yg = (char*)d.y;
  float sc[16], sh[16];
  if constexpr (!BITS) {
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      const float4 a = *(const float4*)(ctab + 8 * C + cl + i), b = *(const float4*)(ctab + 9 * C + cl + i);
      sc[i] = a.x; sc[i + 1] = a.y; sc[i + 2] = a.z; sc[i + 3] = a.w;
      sh[i] = b.x; sh[i + 1] = b.y; sh[i + 2] = b.z; sh[i + 3] = b.w;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float v[8], xv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = acc[t][8 * hh + i];
      if constexpr (RES) {
        float r[8];
        Elem<bf16_t>::unpack(e.rr[t][hh], r);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += row_keep_if_bit(r[i], e.rb[t], 8 * hh + i);
      }
      const uint4 pk = Elem<bf16_t>::pack(v);
      if ((okmask >> t) & 1) *(uint4*)(yg + off[t] + 16 * hh) = pk;
      Elem<bf16_t>::unpack(pk, v);            // the values as stored
      Elem<bf16_t>::unpack(e.xr[t][hh], xv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float g;
        if constexpr (BITS) g = row_keep_if_bit(v[i], e.mb[t], 8 * hh + i);
        else g = row_bn_act(xv[i], sc[8 * hh + i], sh[8 * hh + i]) > 0.f ? v[i] : 0.f;
        s1[8 * hh + i] += g;
        s2[8 * hh + i] = fmaf(g, xv[i], s2[8 * hh + i]);
      }
    }
  }
}

template <int NT, bool BITS, bool RES>
__device__ __forceinline__ void row_epi_bnb(const hrp_conv_desc& d, const f32x16 (&acc)[NT], const unsigned (&off)[NT],
                                            const unsigned okmask, const int cl, const float* ctab, const int C,
                                            float (&s1)[16], float (&s2)[16]) {
  RowEpiOps<NT> e;
  row_epi_load<NT, true, BITS, RES>(d, off, okmask, e);
  row_epi_bnb_math<NT, BITS, RES>(d, acc, off, okmask, cl, ctab, C, e, s1, s2);
}

template <int NT, bool EXT, bool STATS = true>
__device__ __forceinline__ void row_epilogue(const hrp_conv_desc& d, const f32x16 (&acc)[NT], const unsigned (&off)[NT],
                                             const unsigned okmask, const int cl, const float* ctab, const int C, const bool bnb,
                                             float (&s1)[16], float (&s2)[16]) {
  char* yg = (char*)d.y;
  if (bnb) {
    // (workgroup-uniform choices - descriptor pointers - select the code version with scalar branches)
    if constexpr (!EXT) row_epi_bnb<NT, false, false>(d, acc, off, okmask, cl, ctab, C, s1, s2);
    else {
      const bool ubits = d.bnb_mask != nullptr, ures = d.res != nullptr;
      if (ubits) { if (ures) row_epi_bnb<NT, true, true>(d, acc, off, okmask, cl, ctab, C, s1, s2); else row_epi_bnb<NT, true, false>(d, acc, off, okmask, cl, ctab, C, s1, s2); }
      else { if (ures) row_epi_bnb<NT, false, true>(d, acc, off, okmask, cl, ctab, C, s1, s2); else row_epi_bnb<NT, false, false>(d, acc, off, okmask, cl, ctab, C, s1, s2); }
    }
    return;
  }
  const char* rg_ = (const char*)d.res;
  uint4 rr[NT][2];
  if (rg_) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      rr[t][0] = rr[t][1] = make_uint4(0, 0, 0, 0);
      if ((okmask >> t) & 1) { rr[t][0] = *(const uint4*)(rg_ + off[t]); rr[t][1] = *(const uint4*)(rg_ + off[t] + 16); }
    }
  }
  float sc[16], sh[16];
  const bool aff = d.scale != nullptr;
  if (aff) {
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      const float4 a = *(const float4*)(d.scale + cl + i), b = *(const float4*)(d.shift + cl + i);
      sc[i] = a.x; sc[i + 1] = a.y; sc[i + 2] = a.z; sc[i + 3] = a.w;
      sh[i] = b.x; sh[i + 1] = b.y; sh[i + 2] = b.z; sh[i + 3] = b.w;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = acc[t][8 * hh + i];
      if (aff) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[8 * hh + i] + sh[8 * hh + i];
      }
      if (rg_) {
        float r[8];
        Elem<bf16_t>::unpack(rr[t][hh], r);
        if constexpr (EXT) {
          if (d.res_mask && ((okmask >> t) & 1)) {
            const int rb = d.res_mask[(off[t] >> 4) + hh];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = row_keep_if_bit(r[i], rb, i);
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += r[i];
      }
      if (d.relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
      }
      const uint4 pk = Elem<bf16_t>::pack(v);
      if ((okmask >> t) & 1) *(uint4*)(yg + off[t] + 16 * hh) = pk;
      if (STATS && d.stats && ((okmask >> t) & 1)) {
        Elem<bf16_t>::unpack(pk, v);          // statistics of the values as stored
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[8 * hh + i] += v[i]; s2[8 * hh + i] = fmaf(v[i], v[i], s2[8 * hh + i]); }
      }
    }
  }
}

// Reduce-scatter over the 32 lanes of a half wave (they hold sums of the same 16 channels): after the step with lane
// distance D a lane keeps the half of its values selected by bit D of its lane id; 31 shuffles instead of 32 x 5.
// -> lane l31 < 16: the total of s1[l31]; l31 >= 16: the total of s2[l31 - 16].
__device__ __forceinline__ float row_reduce32(const float (&s1)[16], const float (&s2)[16], const int l31) {
  float v16[16], v8[8], v4[4], v2[2];
  {
    const bool up = (l31 & 16) != 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float keep = up ? s2[j] : s1[j], send = up ? s1[j] : s2[j]; v16[j] = keep + __shfl_xor(send, 16, 64); }
  }
  {
    const bool up = (l31 & 8) != 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float keep = up ? v16[8 + j] : v16[j], send = up ? v16[j] : v16[8 + j]; v8[j] = keep + __shfl_xor(send, 8, 64); }
  }
  {
    const bool up = (l31 & 4) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float keep = up ? v8[4 + j] : v8[j], send = up ? v8[j] : v8[4 + j]; v4[j] = keep + __shfl_xor(send, 4, 64); }
  }
  {
    const bool up = (l31 & 2) != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) { const float keep = up ? v4[2 + j] : v4[j], send = up ? v4[j] : v4[2 + j]; v2[j] = keep + __shfl_xor(send, 2, 64); }
  }
  const bool up = (l31 & 1) != 0;
  const float keep = up ? v2[1] : v2[0], send = up ? v2[0] : v2[1];
  return keep + __shfl_xor(send, 1, 64);
}

// a wave that owns its 32 channels (deep kernels): straight to the statistic slot
__device__ __forceinline__ void row_stats_commit(const hrp_conv_desc& d, const float v1, const int l31, const int cl, const float* ctab,
                                                 const int C, const bool bnb, const int stat_slot) {
  const int which = l31 >> 4, c = cl + (l31 & 15);
  const float other = __shfl_xor(v1, 16, 64);     // sum 1 of the same channel, for the lanes holding sum 2
  float tot = v1;
  if (bnb && which == 1) tot = fmaf(ctab[6 * C + c], v1, ctab[7 * C + c] * other);     // sum g * xhat = a * sum g x + b * sum g
  atomicAdd(d.stats + stat_slot * 2 * C + which * C + c, (double)tot);
}

template <int C, bool EXT>
__device__ __forceinline__ void conv_row_body_t(const hrp_conv_desc& d, const RowPlan& rp, int bid, const int stat_slot) {
  using R = RowCfg<C>;
  constexpr int W = R::W, P = R::P, S = R::S, KS = R::KS, TH = R::TH, ROWB = R::ROWB, NROWS = R::NROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ctab = (float*)(smem + R::CTAB_OFF);
  float* stat_lds = (float*)(smem + R::STAT_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int sel = wave & 1, rg = wave >> 1;
  const int col = R::NCOL == 2 ? sel : 0;     // which 32-pixel column block of the rows
  const int m = R::MT == 2 ? sel : 0;         // which 32-channel output tile
  HRP_CSTAMP(0);

  // The workgroup is PERSISTENT over rp.spw consecutive strips (of one image when spw divides the strips per image): the
  // weights, the BatchNorm constants and the zero pixels are set up once, the statistics leave once, and the DMA of strip
  // s + 1 is issued before the epilogue (stores) of strip s.
  const int ngroups = (rp.nstrips + rp.spw - 1) / rp.spw;
  if ((ngroups & 7) == 0) bid = (bid & 7) * (ngroups >> 3) + (bid >> 3);   // neighbouring groups (one image) on one XCD
  const int s_begin = bid * rp.spw, s_end = min(s_begin + rp.spw, rp.nstrips);
  const int H = d.H;
  const int pro = d.pro_mode;
  const bool ub = EXT && d.pro_mask != nullptr, wgm = EXT && d.pro_side2 != nullptr;     // (uniform: RowPro::bwd)

  // ---- weights: A fragments of this wave's 32 output channels.  MFMA row rho = 8 q + 4 h + i carries output channel
  // 16 h + 4 q + i, so that accumulator register 4 q + i of a lane (half h) is channel 16 h + 4 q + i: consecutive.
  // C = 64: 144 registers, loaded once per workgroup (one strip).  C = 32: all four waves use the same 18 fragments: they
  // are DMA'd ONCE into LDS (4.5 pieces per wave instead of 18 global loads per wave, whose issue alone was ~1 us of every
  // workgroup) and re-read into registers for every strip's MFMA loop, so that they do not occupy registers during the
  // prologue / epilogue phases of a persistent workgroup.
  bf16x8 wf[9][KS];
  const int co_lane = m * 32 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  const char* wl = (const char*)d.w + co_lane * ROW + half * 16;
  char* wlds = smem + R::WLDS_OFF;
  if constexpr (R::PERSIST) {
#pragma unroll
    for (int f = 0; f < 9 * KS; ++f)
      if ((f & 3) == wave) dma16(wl + (size_t)(((f % KS) * d.w_ntaps + rp.wslot[f / KS]) * C) * ROW, wlds + f * 1024);
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
        wf[t][kk] = *(const bf16x8*)(wl + (size_t)((kk * d.w_ntaps + rp.wslot[t]) * C) * ROW);
  }

  // ---- staging: piece `wave` of every row; the lane's 16 bytes = (pixel lane / S of the piece, slot lane % S), holding
  // the LOGICAL slot (lane % S) ^ g(x)
  const int px_in_piece = lane / S;
  const int xcol = wave * R::PXP + px_in_piece;
  const int lslot = (lane % S) ^ R::g(xcol);
  const unsigned lane_off = (unsigned)(wave * 1024 + px_in_piece * P + lslot * 16);
  char* lds_rows = smem + P;
  auto strip_of = [&](int s, int& n, int& y0) { n = fdiv(s, rp.fd_spi); y0 = (s - n * rp.spi) * TH; };
  auto stage = [&](int s) {
    int n, y0;
    strip_of(s, n, y0);
    const char* xg = (const char*)d.x + (unsigned)n * (unsigned)(H * W * P) + lane_off;
#pragma unroll
    for (int rs = 0; rs < NROWS; ++rs) {
      const int y = y0 - 1 + rs;
      char* dst = lds_rows + rs * ROWB + wave * 1024;
      if (y >= 0 && y < H) dma16(xg + y * (W * P), dst);
      else *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
    }
  };
  stage(s_begin);
  // the zero pixels: one in front of row slot 0, one behind every row slot
  if (tid < (NROWS + 1) * S) {
    const int k = tid / S, j = tid - k * S;
    *(uint4*)(smem + (k == 0 ? 0 : P + (k - 1) * ROWB + W * P) + j * 16) = make_uint4(0, 0, 0, 0);
  }

  // ---- per-channel constants (LDS table [10][C]):
  //   0 sc, 1 sh        of pro_stats  (prologue 1 / 2: act = fma(x, sc, sh))
  //   2 a = invstd, 3 b = -mean * invstd, 4 k0, 5 k1     (prologue 2)
  //   6 a, 7 b, 8 sc, 9 sh   of bnb_stats (epilogue reduce)
  const bool bnb = d.bnb_x != nullptr;
  if (pro != 0 && tid < C) {
    float mean, inv, sc, sh;
    row_bn_consts(d.pro_stats, d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps, tid, C, mean, inv, sc, sh);
    ctab[0 * C + tid] = sc; ctab[1 * C + tid] = sh;
    if (pro == 2) {
      ctab[2 * C + tid] = inv; ctab[3 * C + tid] = -mean * inv;
      ctab[4 * C + tid] = slot_sum(d.pro_bsums, tid, 2 * C) / d.pro_count;
      ctab[5 * C + tid] = slot_sum(d.pro_bsums, C + tid, 2 * C) / d.pro_count;
    }
  }
  if (bnb && tid >= 64 && tid < 64 + C) {
    const int c = tid - 64;
    float mean, inv, sc, sh;
    row_bn_consts(d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps, c, C, mean, inv, sc, sh);
    ctab[8 * C + c] = sc; ctab[9 * C + c] = sh;
    ctab[6 * C + c] = inv; ctab[7 * C + c] = -mean * inv;
  }
  HRP_CSTAMP(1);
  if (pro != 0) __syncthreads();                                          // the constant table

  // read address of (dx, kk) for row slot rg*4: pixel x = col*32 + l31 + dx (x = -1 / W are the shared zero pixels)
  int baddr[3][KS];
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi) {
    const int xq = col * 32 + l31 + dxi - 1;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      baddr[dxi][kk] = P + rg * 4 * ROWB + xq * P + (((2 * kk + half) ^ R::g(xq)) << 4);
  }
  const int cl = m * 32 + 16 * half;      // first output channel of the lane
  float vtot = 0.f;                       // statistics of the strips so far, already reduced over the half wave

  const int s_stop = R::PERSIST ? s_end : s_begin + 1;        // (C = 64: exactly one strip, known to the compiler)
  for (int s = s_begin; s < s_stop; ++s) {
    int n, y0;
    strip_of(s, n, y0);
    const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this lane's DMA pieces of strip s have landed
    // ---- prologue: transform the bytes this lane staged, in place
    if (pro != 0) {
      RowPro pc;
      int cb = lslot * 8;                                       // the lane's 8 channels
      asm volatile("" : "+v"(cb));                              // (opaque per strip: the constants are re-read from LDS, not kept
      pc.load(ctab, C, cb);                                     //  in registers across the MFMA loop of a persistent workgroup)
      char* side = (char*)d.pro_side;
      if (pro == 1) {
#pragma unroll
        for (int rs = 0; rs < NROWS; ++rs) {
          const int y = y0 - 1 + rs;
          if (y < 0 || y >= H) continue;
          char* p = lds_rows + rs * ROWB + wave * 1024 + lane * 16;
          const uint4 o = pc.act(*(const uint4*)p);
          *(uint4*)p = o;
          if (side && rs >= 1 && rs <= TH) *(uint4*)(side + img_off + lane_off + y * (W * P)) = o;
        }
      } else {
        const bool f3 = EXT && pro == 3;                          // (uniform) block-end forward of the previous block
        if (!f3) pc.load2(ctab, C, cb);
        // second operand: the same bytes of the BatchNorm input (and the mask byte of the vector), through registers,
        // five rows at a time
        const char* x2g = (const char*)d.pro_x2 + img_off + lane_off;
        const uint8_t* mg = (EXT && d.pro_mask) ? d.pro_mask + ((img_off + lane_off) >> 4) : nullptr;
#pragma unroll
        for (int r0 = 0; r0 < NROWS; r0 += 5) {
          uint4 x2[5];
          int bits[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int y = y0 - 1 + r0 + j;
            x2[j] = make_uint4(0, 0, 0, 0);
            bits[j] = -1;
            if (y >= 0 && y < H) {
              x2[j] = *(const uint4*)(x2g + y * (W * P));
              if constexpr (EXT) {
                if (mg && !f3) bits[j] = mg[y * (W * P / 16)];
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int rs = r0 + j, y = y0 - 1 + rs;
            if (y < 0 || y >= H) continue;
            char* p = lds_rows + rs * ROWB + wave * 1024 + lane * 16;
            if constexpr (EXT) {
              if (f3) {
                unsigned ob;
                const uint4 o3 = pc.fwd3(*(const uint4*)p, x2[j], ob);
                *(uint4*)p = o3;
                if (rs >= 1 && rs <= TH) {
                  *(uint4*)(side + img_off + lane_off + y * (W * P)) = o3;
                  const_cast<uint8_t*>(mg)[y * (W * P / 16)] = (uint8_t)ob;
                }
                continue;
              }
            }
            uint4 gm;
            const uint4 o = pc.template bwd<EXT>(*(const uint4*)p, x2[j], bits[j], gm, ub, wgm);
            *(uint4*)p = o;
            if (rs >= 1 && rs <= TH) {
              const unsigned off = img_off + lane_off + y * (W * P);
              if (side) *(uint4*)(side + off) = o;
              if constexpr (EXT) {
                if (d.pro_side2) row_side2(d, off, gm);
              }
            }
          }
        }
      }
    }
    __syncthreads();
    if (s == s_begin) HRP_CSTAMP(2);
    if constexpr (R::PERSIST) {
#pragma unroll
      for (int f = 0; f < 9 * KS; ++f) wf[f / KS][f % KS] = *(const bf16x8*)(wlds + f * 1024 + lane * 16);
    }

    // ---- MFMA loop: input rows rg*4 - 1 .. rg*4 + 4 of the strip (LDS row slots rg*4 .. rg*4 + 5)
    f32x16 acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[o][i] = 0.f;
    {
      constexpr int NSTEP = 6 * 3 * KS, RING = 4, AHEAD = 3;
      bf16x8 bq[RING];
      auto rd = [&](int q) -> bf16x8 {   // q is a constant after unrolling
        const int irel = q / (3 * KS), dxi = (q / KS) % 3, kk = q % KS;
        return *(const bf16x8*)(smem + baddr[dxi][kk] + irel * ROWB);
      };
#pragma unroll
      for (int q = 0; q < AHEAD; ++q) bq[q % RING] = rd(q);
#pragma unroll
      for (int q = 0; q < NSTEP; ++q) {
        if (q + AHEAD < NSTEP) bq[(q + AHEAD) % RING] = rd(q + AHEAD);
        const int irel = q / (3 * KS), dxi = (q / KS) % 3, kk = q % KS;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int dyi = irel - o;                 // input row (rg*4 - 1 + irel) = output row (rg*4 + o) + dy, dy = dyi - 1
          if (dyi >= 0 && dyi <= 2)
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[dyi * 3 + dxi][kk], bq[q % RING], acc[o], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the read-ahead where it is
      }
    }
    if (s == s_begin) HRP_CSTAMP(4);
    // every wave is done with the rows: the next strip's DMA runs under this strip's epilogue
    if constexpr (R::PERSIST) {
      if (s + 1 < s_end) {
        __syncthreads();
        stage(s + 1);
      }
    }

    // ---- epilogue: lane = pixel (row y0 + rg*4 + o, x = col*32 + l31), channels m*32 + 16*half .. +15
    float s1[16], s2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
    {
      const unsigned out_off = img_off + (unsigned)((y0 + rg * 4) * W + col * 32 + l31) * P + cl * 2;
      constexpr int EG = EXT ? 2 : 4;          // tiles per epilogue group: bounds the registers of the residual / bnb_x rows
#pragma unroll
      for (int o0 = 0; o0 < 4; o0 += EG) {
        unsigned off[EG];
#pragma unroll
        for (int o = 0; o < EG; ++o) off[o] = out_off + (o0 + o) * (W * P);
        if constexpr (EG == 4) row_epilogue<4, EXT>(d, acc, off, 0xfu, cl, ctab, C, bnb, s1, s2);
        else row_epilogue<EG, EXT>(d, *(const f32x16(*)[EG])&acc[o0], off, (1u << EG) - 1, cl, ctab, C, bnb, s1, s2);
      }
    }
    if (d.stats) vtot += row_reduce32(s1, s2, l31);
    if (s == s_begin) HRP_CSTAMP(5);
  }
  if (d.stats) {
    // lane l31 < 16: sum 1 of channel cl + l31; l31 >= 16: sum 2 of channel cl + l31 - 16
    stat_lds[wave * 64 + lane] = vtot;
    __syncthreads();
    if (tid < 2 * C) {
      const int which = tid / C, c = tid - which * C;
      const int mc = c >> 5, hq = (c >> 4) & 1, j = c & 15;
      auto tot = [&](int wh) {
        const int li = hq * 32 + wh * 16 + j;
        float t = 0.f;
        if (R::MT == 1) t = (stat_lds[0 * 64 + li] + stat_lds[1 * 64 + li]) + (stat_lds[2 * 64 + li] + stat_lds[3 * 64 + li]);
        else t = stat_lds[mc * 64 + li] + stat_lds[(mc + 2) * 64 + li];
        return t;
      };
      float t = tot(which);
      if (bnb && which == 1) t = fmaf(ctab[6 * C + c], t, ctab[7 * C + c] * tot(0));     // sum g * xhat = a * sum g x + b * sum g
      atomicAdd(d.stats + stat_slot * 2 * C + which * C + c, (double)t);
    }
  }
  HRP_CSTAMP(6);
  HRP_CSTAMP(7);
}

template <int C>
__device__ __forceinline__ void conv_row_body(const hrp_conv_desc& d, const RowPlan& rp, const int bid, const int stat_slot) {
  if (row_ext(d)) conv_row_body_t<C, true>(d, rp, bid, stat_slot);
  else conv_row_body_t<C, false>(d, rp, bid, stat_slot);
}

template <int C>
__global__ __launch_bounds__(256, 2) void conv_row_kernel(const hrp_conv_desc d, const RowPlan rp) {
  conv_row_body<C>(d, rp, blockIdx.x, blockIdx.x & (HRP_STAT_SLOTS - 1));
}


// =====================================================================================================================
// Deep variant: C = 128 @ W = 16 and C = 256 @ W = 8 (the low-resolution branches; K = 9 C = 1 152 / 2 304).  Same skeleton
// - input rows resident in LDS with ALL their channels, staged by lane-affine DMA pieces, BatchNorm prologues in place,
// epilogue straight from registers - but the weights no longer fit a wave's registers: they STREAM through them, one
// 16-channel K chunk (9 taps x MW fragments) per step, double buffered, each fragment one coalesced 1 KiB global load
// of the packed layout (L2 resident: every workgroup of a layer reads the same slab at about the same time).  No weight
// ever touches LDS and no wave issues an LDS-DMA piece inside the MFMA loop (the general tile program spends ~100 cycles
// of a wave's issue slot per piece there, which is why its deep-K loops run the matrix pipe at ~50 %).
//   C = 128: workgroup = 8 rows x 16 px of one image x 128 channels out; wave = 32 channels out x 128 px (MW 1, NW 4)
//   C = 256: workgroup = 8 rows x  8 px of one image x 256 channels out; wave = 64 channels out x  64 px (MW 2, NW 2)
// LDS swizzle: a pixel has 16 / 32 slots; slot' = slot ^ f with f = (y W + x) & 15 of the source pixel's IMAGE position
// (distinct over the 16 lanes of a ds_read_b128 group, whose pixels are 16 consecutive positions of a 32-pixel tile).
template <int C>
struct DeepCfg {
  static constexpr int W = 2048 / C;        // 16, 8
  static constexpr int P = 2 * C;           // 256, 512 bytes per pixel
  static constexpr int S = P / 16;          // 16, 32 slots per pixel
  static constexpr int KS = C / 16;         // 8, 16 K chunks
  static constexpr int PXP = 1024 / P;      // 4, 2 pixels per DMA piece
  static constexpr int ROWB = (W + 1) * P;
  static constexpr int TH = 8, NROWS = TH + 2;
  static constexpr int MW = C == 128 ? 1 : 2, NW = 4 / MW;     // wave tile: MW x NW blocks of 32 x 32
  static constexpr int RPT = 32 / W;        // image rows per 32-pixel tile: 2, 4
  static constexpr int TILE_BYTES = P + NROWS * ROWB;
  static constexpr int CTAB_OFF = (TILE_BYTES + 255) & ~255;
  static constexpr int LDS_BYTES = CTAB_OFF + 10 * C * 4;
  static_assert(NW * RPT == TH, "the wave's tiles cover the strip");
  __device__ static __forceinline__ int f(int y, int x) { return (y * W + x) & 15; }
};

template <int C, bool EXT>
__device__ __forceinline__ void conv_deep_body_t(const hrp_conv_desc& d, const RowPlan& rp, int bid, const int stat_slot) {
  using R = DeepCfg<C>;
  constexpr int W = R::W, P = R::P, S = R::S, KS = R::KS, TH = R::TH, ROWB = R::ROWB, NROWS = R::NROWS;
  constexpr int MW = R::MW, NW = R::NW, RPT = R::RPT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ctab = (float*)(smem + R::CTAB_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  HRP_CSTAMP(0);
  if ((rp.nstrips & 7) == 0) bid = (bid & 7) * (rp.nstrips >> 3) + (bid >> 3);
  const int n = fdiv(bid, rp.fd_spi);
  const int y0 = (bid - n * rp.spi) * TH;
  const int H = d.H;
  const int pro = d.pro_mode;
  const bool ub = EXT && d.pro_mask != nullptr, wgm = EXT && d.pro_side2 != nullptr;     // (uniform: RowPro::bwd)
  const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
  char* lds_rows = smem + P;

  // ---- staging: piece `wave` of every row (a row is 4 pieces).  lane = (pixel lane / S of the piece, slot lane % S)
  const int px_in_piece = lane / S, pslot = lane % S;
  const int xcol = wave * R::PXP + px_in_piece;
  auto lane_off_of = [&](int y) { return (unsigned)(wave * 1024 + px_in_piece * P + ((pslot ^ R::f(y, xcol)) << 4)); };
  {
    const char* xg = (const char*)d.x + img_off;
#pragma unroll
    for (int rs = 0; rs < NROWS; ++rs) {
      const int y = y0 - 1 + rs;
      char* dst = lds_rows + rs * ROWB + wave * 1024;
      if (y >= 0 && y < H) dma16(xg + y * (W * P) + lane_off_of(y), dst);
      else *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  if (tid < (NROWS + 1) * S) {   // the zero pixels (C = 256: 11 x 32 slots > 256 threads: two rounds)
    for (int e = tid; e < (NROWS + 1) * S; e += 256) {
      const int k = e / S, j = e - k * S;
      *(uint4*)(smem + (k == 0 ? 0 : P + (k - 1) * ROWB + W * P) + j * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  const bool bnb = d.bnb_x != nullptr;
  if (pro != 0) {
    for (int c = tid; c < C; c += 256) {
      float mean, inv, sc, sh;
      row_bn_consts(d.pro_stats, d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps, c, C, mean, inv, sc, sh);
      ctab[0 * C + c] = sc; ctab[1 * C + c] = sh;
      if (pro == 2) {
        ctab[2 * C + c] = inv; ctab[3 * C + c] = -mean * inv;
        ctab[4 * C + c] = slot_sum(d.pro_bsums, c, 2 * C) / d.pro_count;
        ctab[5 * C + c] = slot_sum(d.pro_bsums, C + c, 2 * C) / d.pro_count;
      }
    }
  }
  if (bnb) {
    for (int c = tid; c < C; c += 256) {
      float mean, inv, sc, sh;
      row_bn_consts(d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps, c, C, mean, inv, sc, sh);
      ctab[8 * C + c] = sc; ctab[9 * C + c] = sh;
      ctab[6 * C + c] = inv; ctab[7 * C + c] = -mean * inv;
    }
  }
  HRP_CSTAMP(1);

  // ---- prologue: the lane's channels depend on the row through the swizzle only for C = 256 (two row parities)
  if (pro != 0) {
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    char* side = (char*)d.pro_side;
    constexpr int NPAR = C == 256 ? 2 : 1;
#pragma unroll
    for (int par = 0; par < NPAR; ++par) {
      RowPro pc;
      const int cb = (pslot ^ R::f(par, xcol)) * 8;               // f depends on y only through its parity (W = 8) or not at all
      pc.load(ctab, C, cb);
      if (pro == 1) {
#pragma unroll
        for (int rs = 0; rs < NROWS; ++rs) {
          const int y = y0 - 1 + rs;
          if (y < 0 || y >= H || (NPAR == 2 && ((y & 1) != par))) continue;
          char* p = lds_rows + rs * ROWB + wave * 1024 + lane * 16;
          const uint4 o = pc.act(*(const uint4*)p);
          *(uint4*)p = o;
          if (side && rs >= 1 && rs <= TH) *(uint4*)(side + img_off + y * (W * P) + lane_off_of(y)) = o;
        }
      } else {
        const bool f3 = EXT && pro == 3;                          // (uniform) block-end forward of the previous block
        if (!f3) pc.load2(ctab, C, cb);
        // the BatchNorm inputs (and mask bytes) of this parity's rows: same lane-constant addressing, through registers
        uint4 x2[NROWS];
        int bits[NROWS];
#pragma unroll
        for (int rs = 0; rs < NROWS; ++rs) {
          const int y = y0 - 1 + rs;
          x2[rs] = make_uint4(0, 0, 0, 0);
          bits[rs] = -1;
          if (y >= 0 && y < H && !(NPAR == 2 && ((y & 1) != par))) {
            const unsigned off = img_off + y * (W * P) + lane_off_of(y);
            x2[rs] = *(const uint4*)((const char*)d.pro_x2 + off);
            if constexpr (EXT) {
              if (d.pro_mask && !f3) bits[rs] = d.pro_mask[off >> 4];
            }
          }
        }
#pragma unroll
        for (int rs = 0; rs < NROWS; ++rs) {
          const int y = y0 - 1 + rs;
          if (y < 0 || y >= H || (NPAR == 2 && ((y & 1) != par))) continue;
          char* p = lds_rows + rs * ROWB + wave * 1024 + lane * 16;
          if constexpr (EXT) {
            if (f3) {
              unsigned ob;
              const uint4 o3 = pc.fwd3(*(const uint4*)p, x2[rs], ob);
              *(uint4*)p = o3;
              if (rs >= 1 && rs <= TH) {
                const unsigned off = img_off + y * (W * P) + lane_off_of(y);
                *(uint4*)(side + off) = o3;
                const_cast<uint8_t*>(d.pro_mask)[off >> 4] = (uint8_t)ob;
              }
              continue;
            }
          }
          uint4 gm;
          const uint4 o = pc.template bwd<EXT>(*(const uint4*)p, x2[rs], bits[rs], gm, ub, wgm);
          *(uint4*)p = o;
          if (rs >= 1 && rs <= TH) {
            const unsigned off = img_off + y * (W * P) + lane_off_of(y);
            if (side) *(uint4*)(side + off) = o;
            if constexpr (EXT) {
              if (d.pro_side2) row_side2(d, off, gm);
            }
          }
        }
      }
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  HRP_CSTAMP(2);

  // ---- MFMA loop.  Tile t of the wave: strip rows t*RPT .. +RPT-1; lane = pixel (row l31 / W, x = l31 % W).
  f32x16 acc[MW][NW];
#pragma unroll
  for (int mi = 0; mi < MW; ++mi)
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][t][i] = 0.f;
  {
    const int r = l31 / W, x = l31 % W;
    // weight rows of the lane (MFMA row rho = l31 carries channel 16 h + 4 q + i of the block, see conv_row_body)
    const int co_l = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    const char* wl = (const char*)d.w + (size_t)((wave * MW) * 32 + co_l) * ROW + half * 16;
    auto wload = [&](int kk, bf16x8 (&wb)[MW][9]) {
#pragma unroll
      for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          wb[mi][t] = *(const bf16x8*)(wl + (size_t)((kk * d.w_ntaps + rp.wslot[t]) * C + mi * 32) * ROW);
    };
    // read addresses: corner (dy = -1, dx = -1) of tile t, XOR-ed with the swizzle of the tap's source pixel; the K chunk
    // adds kk << 5 by XOR as well (the pixel base is P-aligned), dy / dx by immediate offsets
    constexpr int NCLS = C == 256 ? 6 : 3;       // swizzle classes of a tap: (row parity of the source,) dx
    int bt[NW][NCLS];
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      const int yr = t * RPT + r;                                   // strip row of the lane's pixel
      const int corner = P + yr * ROWB + (x - 1) * P;               // slot yr = strip row yr - 1: the dy = -1 row
#pragma unroll
      for (int cls = 0; cls < NCLS; ++cls) {
        const int dxi = cls % 3, par = cls / 3;                     // par: parity of (source row - lane row) (C = 256 only)
        const int fy = C == 256 ? ((y0 + yr + par) & 1) : 0;
        bt[t][cls] = corner ^ (((half ^ R::f(fy, x + dxi - 1)) & (S - 1)) << 4);
      }
    }
    constexpr int CH = 9 * NW, RING = MW == 1 ? 6 : 3, AHEAD = RING - 1;     // steps of one K chunk (a multiple of RING)
    static_assert(CH % RING == 0 && KS % 2 == 0, "ring positions repeat per chunk; chunks are processed in pairs");
    bf16x8 bq[RING];
    auto rd = [&](int kkoff, int q) -> bf16x8 {      // q = tap * NW + t, a constant after unrolling; kkoff = kk << 5
      const int tap = q / NW, t = q % NW;
      const int dyi = tap / 3, dxi = tap % 3;
      const int cls = C == 256 ? ((dyi + 1) & 1) * 3 + dxi : dxi;   // dy = dyi - 1: odd dy flips the row parity
      return *(const bf16x8*)(smem + (bt[t][cls] ^ kkoff) + dyi * ROWB + dxi * P);
    };
    // one K chunk: 9 taps x NW tiles; the B fragments of step q + AHEAD (into the next chunk at the end) are read while
    // the MFMAs of step q run
    auto chunk = [&](int kk, const bf16x8 (&wb)[MW][9]) {
      const int kkoff = kk << 5;
      const bool more = kk + 1 < KS;
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        if (q + AHEAD < CH) bq[(q + AHEAD) % RING] = rd(kkoff, q + AHEAD);
        else if (more) bq[(q + AHEAD) % RING] = rd(kkoff + 32, q + AHEAD - CH);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
          acc[mi][q % NW] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[mi][q / NW], bq[q % RING], acc[mi][q % NW], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    bf16x8 wa[MW][9], wb2[MW][9];
    wload(0, wa);
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) bq[q % RING] = rd(0, q);
    for (int kk = 0; kk < KS; kk += 2) {
      wload(kk + 1, wb2);                 // lands while chunk kk runs
      chunk(kk, wa);
      if (kk + 2 < KS) wload(kk + 2, wa);
      chunk(kk + 1, wb2);
    }
  }
  HRP_CSTAMP(4);

  // ---- epilogue: lane = pixel (strip row t*RPT + l31 / W, x = l31 % W); channels (wave*MW + mi)*32 + 16*half .. +15
  const unsigned pix_off = img_off + (unsigned)((y0 + l31 / W) * W + (l31 % W)) * P;
#pragma unroll
  for (int mi = 0; mi < MW; ++mi) {
    const int cl = (wave * MW + mi) * 32 + 16 * half;
    unsigned off[NW];
#pragma unroll
    for (int t = 0; t < NW; ++t) off[t] = pix_off + cl * 2 + t * (RPT * W * P);
    float s1[16], s2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
    row_epilogue<NW, EXT>(d, acc[mi], off, (1u << NW) - 1, cl, ctab, C, bnb, s1, s2);
    if (d.stats) row_stats_commit(d, row_reduce32(s1, s2, l31), l31, cl, ctab, C, bnb, stat_slot);
  }
  HRP_CSTAMP(6);
  HRP_CSTAMP(7);
}

template <int C>
__device__ __forceinline__ void conv_deep_body(const hrp_conv_desc& d, const RowPlan& rp, const int bid, const int stat_slot) {
  if (row_ext(d)) conv_deep_body_t<C, true>(d, rp, bid, stat_slot);
  else conv_deep_body_t<C, false>(d, rp, bid, stat_slot);
}

template <int C>
__global__ __launch_bounds__(256, 2) void conv_deep_kernel(const hrp_conv_desc d, const RowPlan rp) {
  conv_deep_body<C>(d, rp, blockIdx.x, blockIdx.x & (HRP_STAT_SLOTS - 1));
}

// =====================================================================================================================
// Whole-image variant of the deep kernels: C = 128 @ 16 x 16 and C = 256 @ 8 x 8, the production shapes of branches 2 / 3.
// The strip kernels above stream every weight fragment for 128 pixels (A fragment : MFMA = 1 : 4 resp. 1 : 2); with every
// CU of an XCD reading the same weight lines at the same time that saturates the XCD's L2 (~1 KiB / clk measured) and the
// MFMA loop runs at ~45 %.  Here a wave owns 32 output channels x 256 pixels (C = 128: one image, 8 tiles, 1 : 8) or
// x 128 pixels (C = 256: two images, 4 tiles, 1 : 4 with half as many workgroups streaming).
//   LDS layout ("tile linear"): tile t = 32 consecutive pixels of the image(s) in memory order (2 rows of 16 / 4 rows of 8)
//   at t * T, T = 33 pixels: 32 pixels + ONE zero pixel.  A tap's source pixel of lane p is p' = p + dy W + dx of the same
//   image: inside the tile, or in the previous / next tile - which, the zero pixels sitting between tiles, is the same
//   affine address -/+ one pixel.  Out-of-image sources (left / right columns, the row above the top tile, below the
//   bottom tile) read the tile's zero pixel.  So every read is  (B[tap] ^ kk << 5) + t * T  with 15 lane constants B
//   (9 taps + 3 top-row + 3 bottom-row overrides) and an immediate t * T: no padding rows, 64 KiB of LDS for 256 / 128
//   pixels of 128 / 256 channels.  The 64 DMA pieces of a workgroup are one contiguous 64 KiB range of the tensor.
template <int C>
struct ImgCfg {
  static constexpr int W = 2048 / C;              // 16, 8 (square images)
  static constexpr int P = 2 * C, S = P / 16, KS = C / 16;
  static constexpr int NT = C == 128 ? 8 : 4;     // 32-pixel tiles per workgroup (= per wave)
  static constexpr int TPI = W * W / 32;          // tiles per image: 8, 2
  static constexpr int NIMG = NT / TPI;           // images per workgroup: 1, 2
  static constexpr int T = 33 * P;                // LDS tile pitch
  static constexpr int PPT = 32 * P / 1024;       // DMA pieces per tile: 8, 16
  static constexpr int PXP = 1024 / P;            // pixels per piece: 4, 2
  static constexpr int NCB = C / 128;             // blocks of 128 output channels (4 waves x 32)
  static constexpr int TILE_BYTES = NT * T;       // 67 584
  static constexpr int CTAB_OFF = TILE_BYTES;
  static constexpr int LDS_BYTES = CTAB_OFF + 10 * C * 4;
  static_assert(TILE_BYTES % 256 == 0 && NT * PPT == 64, "64 pieces of 1 KiB");
};

template <int C, bool EXT>
__device__ __forceinline__ void conv_img_body_t(const hrp_conv_desc& d, const RowPlan& rp, const int bid, const int stat_slot) {
  using R = ImgCfg<C>;
  constexpr int W = R::W, P = R::P, S = R::S, KS = R::KS, NT = R::NT, TPI = R::TPI, T = R::T, PPT = R::PPT, PXP = R::PXP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ctab = (float*)(smem + R::CTAB_OFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  HRP_CSTAMP(0);
  const int cbk = bid % R::NCB, grp = bid / R::NCB;
  const int n0 = grp * R::NIMG;
  const int nvalid = min(R::NIMG, d.N - n0);                 // images of this group inside the batch
  const unsigned okmask = (1u << (nvalid * TPI)) - 1;
  const int pro = d.pro_mode;
  const bool ub = EXT && d.pro_mask != nullptr, wgm = EXT && d.pro_side2 != nullptr;     // (uniform: RowPro::bwd)
  const unsigned grp_off = (unsigned)n0 * (unsigned)(W * W * P);

  // ---- staging: pieces q = wave + 4 i of the group's 64 (one contiguous 64 KiB range).  lane = (pixel lane / S of the
  // piece, slot lane % S); the slot holds logical slot (lane % S) ^ (p'' & 15), p'' = the pixel's index inside its tile
  const int px_in_piece = lane / S, pslot = lane % S;
  auto lslot_of = [&](int i) {      // i: piece index of the wave (its parity matters for C = 256 only)
    const int pq = (wave + 4 * i) % PPT;
    return pslot ^ ((pq * PXP + px_in_piece) & 15);
  };
  auto piece_off = [&](int i) { return (unsigned)((wave + 4 * i) * 1024 + px_in_piece * P + (lslot_of(i) << 4)); };
  {
    const char* xg = (const char*)d.x + grp_off;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = wave + 4 * i, tq = q / PPT;
      char* dst = smem + tq * T + (q % PPT) * 1024;
      if (tq < nvalid * TPI) dma16(xg + piece_off(i), dst);
      else *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  if (tid < NT * S) {   // the zero pixels behind the tiles (128 slots)
    const int k = tid / S, j = tid - k * S;
    *(uint4*)(smem + k * T + 32 * P + j * 16) = make_uint4(0, 0, 0, 0);
  }
  const bool bnb = d.bnb_x != nullptr;
  if (pro != 0) {
    for (int c = tid; c < C; c += 256) {
      float mean, inv, sc, sh;
      row_bn_consts(d.pro_stats, d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps, c, C, mean, inv, sc, sh);
      ctab[0 * C + c] = sc; ctab[1 * C + c] = sh;
      if (pro == 2) {
        ctab[2 * C + c] = inv; ctab[3 * C + c] = -mean * inv;
        ctab[4 * C + c] = slot_sum(d.pro_bsums, c, 2 * C) / d.pro_count;
        ctab[5 * C + c] = slot_sum(d.pro_bsums, C + c, 2 * C) / d.pro_count;
      }
    }
  }
  if (bnb) {
    for (int c = tid; c < C; c += 256) {
      float mean, inv, sc, sh;
      row_bn_consts(d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps, c, C, mean, inv, sc, sh);
      ctab[8 * C + c] = sc; ctab[9 * C + c] = sh;
      ctab[6 * C + c] = inv; ctab[7 * C + c] = -mean * inv;
    }
  }
  HRP_CSTAMP(1);

  // ---- prologue in place (see conv_row_body); the lane's channel set is (pslot ^ swizzle) * 8: one set for C = 128, two
  // (even / odd pieces) for C = 256; the pieces are handled in two halves to bound the registers of the second operand
  if (pro != 0) {
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    char* side = (char*)d.pro_side;
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      // C = 128: pieces i = 8 hsel .. 8 hsel + 7; C = 256: pieces of parity hsel (i = hsel, hsel + 2, ...)
      auto piece_i = [&](int j) { return C == 256 ? 2 * j + hsel : 8 * hsel + j; };
      RowPro pc;
      const int cb = lslot_of(piece_i(0)) * 8;
      pc.load(ctab, C, cb);
      if (pro == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = piece_i(j), q = wave + 4 * i, tq = q / PPT;
          if (tq >= nvalid * TPI) continue;
          char* p = smem + tq * T + (q % PPT) * 1024 + lane * 16;
          const uint4 o = pc.act(*(const uint4*)p);
          *(uint4*)p = o;
          if (side) *(uint4*)(side + grp_off + piece_off(i)) = o;
        }
      } else {
        const bool f3 = EXT && pro == 3;                          // (uniform) block-end forward of the previous block
        if (!f3) pc.load2(ctab, C, cb);
        uint4 x2[8];
        int bits[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = piece_i(j), q = wave + 4 * i;
          x2[j] = make_uint4(0, 0, 0, 0);
          bits[j] = -1;
          if (q / PPT < nvalid * TPI) {
            const unsigned off = grp_off + piece_off(i);
            x2[j] = *(const uint4*)((const char*)d.pro_x2 + off);
            if constexpr (EXT) {
              if (d.pro_mask && !f3) bits[j] = d.pro_mask[off >> 4];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int i = piece_i(j), q = wave + 4 * i, tq = q / PPT;
          if (tq >= nvalid * TPI) continue;
          char* p = smem + tq * T + (q % PPT) * 1024 + lane * 16;
          if constexpr (EXT) {
            if (f3) {
              unsigned ob;
              const uint4 o3 = pc.fwd3(*(const uint4*)p, x2[j], ob);
              *(uint4*)p = o3;
              const unsigned off = grp_off + piece_off(i);
              *(uint4*)(side + off) = o3;
              const_cast<uint8_t*>(d.pro_mask)[off >> 4] = (uint8_t)ob;
              continue;
            }
          }
          uint4 gm;
          const uint4 o = pc.template bwd<EXT>(*(const uint4*)p, x2[j], bits[j], gm, ub, wgm);
          *(uint4*)p = o;
          const unsigned off = grp_off + piece_off(i);
          if (side) *(uint4*)(side + off) = o;
          if constexpr (EXT) {
            if (d.pro_side2) row_side2(d, off, gm);
          }
        }
      }
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  HRP_CSTAMP(2);

  // ---- MFMA loop
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  {
    const int r = l31 / W, x = l31 % W;
    const int co_l = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    const char* wl = (const char*)d.w + (size_t)(cbk * 128 + wave * 32 + co_l) * ROW + half * 16;
    // weights stream through registers in groups of GT taps, NBUF groups in flight (C = 128: the three taps of one
    // kernel row, 3 buffers - 128 accumulator registers leave no room for two 9-tap buffers; C = 256: all 9 taps, 2 buffers)
    constexpr int GT = C == 128 ? 3 : 9, NG = 9 / GT, NBUF = C == 128 ? 3 : 2;
    bf16x8 wbuf[NBUF][GT];
    auto wload = [&](int g, bf16x8 (&wb)[GT], int gi) {      // group g = kk * NG + gi; gi is a constant after unrolling
      const int kk = g / NG;
      if (kk < KS) {
#pragma unroll
        for (int t = 0; t < GT; ++t) wb[t] = *(const bf16x8*)(wl + (size_t)((kk * d.w_ntaps + rp.wslot[gi * GT + t]) * C) * ROW);
      }
    };
    // lane constants of the reads (relative to t * T): Bm = every row of the tile has the source row inside the image;
    // Bt / Bb = the dy = -1 / +1 taps of an image's first / last tile (rows above / below the image read the zero pixel)
    constexpr int RPT = 32 / W;
    auto bconst = [&](int dyi, int dxi, bool row_ok) {
      const int xs = x + dxi - 1;
      const int ps = l31 + (dyi - 1) * W + dxi - 1;             // source pixel relative to the tile
      const int rel = ps * P + (ps < 0 ? -P : ps >= 32 ? P : 0);
      const int v = rel ^ (((half ^ (ps & 15)) & (S - 1)) << 4);
      return (row_ok && xs >= 0 && xs < W) ? v : 32 * P;
    };
    int Bm[9], Bt[3], Bb[3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) Bm[tap] = bconst(tap / 3, tap % 3, true);
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) { Bt[dxi] = bconst(0, dxi, r > 0); Bb[dxi] = bconst(2, dxi, r < RPT - 1); }
#ifndef HRP_IMG_RING
#define HRP_IMG_RING 6
#endif
    constexpr int CH = 9 * NT, RING = HRP_IMG_RING, AHEAD = RING - 1;
    static_assert(CH % RING == 0 && KS % 2 == 0, "ring positions repeat per chunk; chunks are processed in pairs");
    bf16x8 bq[RING];
    auto rd = [&](int kkoff, int q) -> bf16x8 {      // q = tap * NT + t
      const int tap = q / NT, t = q % NT;
      const int dyi = tap / 3, dxi = tap % 3;
      const int b = (dyi == 0 && t % TPI == 0) ? Bt[dxi] : (dyi == 2 && t % TPI == TPI - 1) ? Bb[dxi] : Bm[tap];
      return *(const bf16x8*)(smem + (b ^ kkoff) + t * T);
    };
    // one K chunk (u = its parity: the loop below runs chunks in pairs so that buffer indices are constants)
    auto chunk = [&](int kk, auto uc) {
      constexpr int u = decltype(uc)::value;
      const int kkoff = kk << 5;
      const bool more = kk + 1 < KS;
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        constexpr int GS = GT * NT;                       // steps per weight group
        const int gi = q / GS;
        if (q % GS == 0) {                                // group (kk, gi) starts: fetch the group NBUF - 1 ahead
          const int g = kk * NG + gi;
          if constexpr (C == 128) wload(g + 2, wbuf[(gi + 2) % 3], (gi + 2) % 3);
          else wload(g + 1, wbuf[(u + 1) & 1], 0);
        }
        if (q + AHEAD < CH) bq[(q + AHEAD) % RING] = rd(kkoff, q + AHEAD);
        else if (more) bq[(q + AHEAD) % RING] = rd(kkoff + 32, q + AHEAD - CH);
        const int bi = C == 128 ? gi : u;
        acc[q % NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wbuf[bi][(q / NT) % GT], bq[q % RING], acc[q % NT], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll
    for (int g = 0; g < NBUF - 1; ++g) wload(g, wbuf[g], g % NG);
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) bq[q % RING] = rd(0, q);
    for (int kk = 0; kk < KS; kk += 2) {
      chunk(kk, std::integral_constant<int, 0>{});
      chunk(kk + 1, std::integral_constant<int, 1>{});
    }
  }
  HRP_CSTAMP(4);

  // ---- epilogue: tile t, lane p = pixel n0 * W * W + 32 t + p of the tensor (contiguous); channels cbk*128 + wave*32 + 16*half ..
  {
    const int cl = cbk * 128 + wave * 32 + 16 * half;
    const unsigned pix0 = grp_off + (unsigned)l31 * P + cl * 2;
    float s1[16], s2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
    constexpr int EG = NT == 8 ? 2 : 4;       // tiles per epilogue group: bounds the registers of the residual / bnb_x rows
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += EG) {
      unsigned off[EG];
#pragma unroll
      for (int t = 0; t < EG; ++t) off[t] = pix0 + (unsigned)(t0 + t) * (32 * P);
      row_epilogue<EG, EXT>(d, *(const f32x16(*)[EG])&acc[t0], off, (okmask >> t0) & ((1u << EG) - 1), cl, ctab, C, bnb, s1, s2);
    }
    if (d.stats) row_stats_commit(d, row_reduce32(s1, s2, l31), l31, cl, ctab, C, bnb, stat_slot);
  }
  HRP_CSTAMP(6);
  HRP_CSTAMP(7);
}

template <int C>
__device__ __forceinline__ void conv_img_body(const hrp_conv_desc& d, const RowPlan& rp, const int bid, const int stat_slot) {
  if (row_ext(d)) conv_img_body_t<C, true>(d, rp, bid, stat_slot);
  else conv_img_body_t<C, false>(d, rp, bid, stat_slot);
}

template <int C>
__global__ __launch_bounds__(256, 2) void conv_img_kernel(const hrp_conv_desc d, const RowPlan rp) {
  conv_img_body<C>(d, rp, blockIdx.x, blockIdx.x & (HRP_STAT_SLOTS - 1));
}

// workgroups of a planned problem
static inline int row_grid(const RowPlan& rp, int C) {
  return (C <= 64 && !rp.img) ? (rp.nstrips + rp.spw - 1) / rp.spw : rp.nstrips;
}

static inline int row_lds_bytes(int C, int img) {
  if (img) return C == 128 ? ImgCfg<128>::LDS_BYTES : ImgCfg<256>::LDS_BYTES;
  return C == 32 ? RowCfg<32>::LDS_BYTES : C == 64 ? RowCfg<64>::LDS_BYTES : C == 128 ? DeepCfg<128>::LDS_BYTES : DeepCfg<256>::LDS_BYTES;
}

// -> HRP_OK when launched, -100 when the problem is not a row-strip problem
static int launch_conv_row(const hrp_conv_desc& d, hipStream_t s) {
  const int C = row_channels(d);
  if (!C) return -100;
  RowPlan rp;
  row_plan(d, rp);
  // a 128-channel layer launched ALONE (inference plans: the third branch's convolutions have no batch partner) takes the
  // half-image variant: 2 N workgroups instead of N - at B = 64 the whole-image kernel occupies a quarter of the chip
  // (forward_only 7.255 -> 7.17 ms).  Inside a batch the whole-image form stays (the other problems fill the chip).
  if (rp.img && C == 128) { rp.img = 0; rp.nstrips = d.N * rp.spi; }
  if (C == 32) hipLaunchKernelGGL(conv_row_kernel<32>, dim3(row_grid(rp, 32)), dim3(256), RowCfg<32>::LDS_BYTES, s, d, rp);
  else if (C == 64) hipLaunchKernelGGL(conv_row_kernel<64>, dim3(row_grid(rp, 64)), dim3(256), RowCfg<64>::LDS_BYTES, s, d, rp);
  else if (rp.img) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)conv_img_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_img_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    if (C == 128) hipLaunchKernelGGL(conv_img_kernel<128>, dim3(rp.nstrips), dim3(256), ImgCfg<128>::LDS_BYTES, s, d, rp);
    else hipLaunchKernelGGL(conv_img_kernel<256>, dim3(rp.nstrips), dim3(256), ImgCfg<256>::LDS_BYTES, s, d, rp);
  }
  else if (C == 128) hipLaunchKernelGGL(conv_deep_kernel<128>, dim3(rp.nstrips), dim3(256), DeepCfg<128>::LDS_BYTES, s, d, rp);
  else {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv_deep_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(conv_deep_kernel<256>, dim3(rp.nstrips), dim3(256), DeepCfg<256>::LDS_BYTES, s, d, rp);
  }
  return check_launch("conv_row_kernel");
}

}  // namespace hrp
