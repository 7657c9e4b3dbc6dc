// Pointwise (1x1, stride 1) convolution for the wide high-resolution layers of HRNet (gfx950, bf16): the Bottleneck 1x1
// layers of layer1 and of the classification-style head (reference HRnet.py:60-98: conv1 256 -> 64, conv3 64 -> 256 at
// 64 x 64), their data gradients, and the other dense 1x1 layers with >= PW_MIN_PIXELS output pixels.
//
// Why a third kernel: a 1x1 convolution with 32 .. 256 input channels has 2 .. 16 MFMA k-steps per output tile - the general
// tile program (conv_tile.h) never leaves its pipeline prologue, stages everything through LDS and round-trips the output
// through LDS again: 64 -> 256 @ 64 x 64 x 64 images ran at 1.7 TB/s of algorithmic traffic.  A 1x1 convolution needs no
// halo, so nothing has to be staged at all:
//   B operand : lane (pixel l & 31, half l >> 5) of MFMA 32x32x16 needs input channels 16 kk + 8 half .. + 7 of its pixel:
//               16 contiguous bytes of the NHWC tensor, read STRAIGHT from global memory into the operand registers (the KS
//               loads of a tile touch every 128-byte line of its 32 pixels exactly once between them)
//   A operand : the wave's 32 MW output channels x all input channels live in registers (KS x MW fragments), read once per
//               workgroup from the packed layout of hrp_pack_weights; MFMA row -> channel permuted as in conv_row.h so that
//               a lane's 16 accumulators are 16 consecutive channels of its pixel (two 16-byte stores, no LDS)
//   workgroup : 4 waves = WC (channel blocks of 32 MW) x WP (pixel tiles); it walks a contiguous run of 32-pixel tiles, the
//               next tile's operand loads in flight under the current tile's MFMAs and stores (PF)
//   statistics: from a second, transposed product per tile (lane = one channel x 16 pixels: sums without cross-lane traffic),
//               accumulated over the tiles of the workgroup; one atomic per channel per wave at the end
// Epilogue options: the ones of conv_row.h's row_epilogue (affine, residual, ReLU, statistics; the BatchNorm-backward reduce
// with the ReLU mask as bits or recomputed, constants from bnb_consts or from bnb_stats).  No bias, no prologue.
#pragma once
#include "conv_row.h"

namespace hrp {

constexpr long PW_MIN_PIXELS = 131072;     // below: the tile program (those layers sit in lock-step batches of small problems)

struct PwPlan {
  int ks, mw;          // k-steps (Cin / 16), channel blocks per wave: the kernel instantiation
  int wc, wp;          // waves across channel blocks / across pixel tiles (wc * wp == 4)
  int groups;          // channel groups of 32 * mw * wc output channels (every group reads x once)
  int ntiles, tpw;     // 32-pixel tiles, tiles per wave (contiguous run of tpw * wp tiles per workgroup)
  int wgs;             // workgroups per channel group
};

// Host: is this problem one the pointwise kernel takes?  -> 1 and the plan, else 0.
static inline int pw_plan(const hrp_conv_desc& d, PwPlan& p) {
  const char* mp = getenv("HRP_PW_MIN_PIXELS");                 // (read per call: tests lower the threshold for small problems, or raise
  const long min_px = mp ? atol(mp) : PW_MIN_PIXELS;            // it beyond every problem to compare with the general tile program)
  if (d.dtype != HRP_BF16 || d.ntaps != 1 || d.in_stride != 1 || d.out_stride != 1) return 0;
  if (d.dy[0] != 0 || d.dx[0] != 0 || d.H != d.Ho || d.W != d.Wo || d.y_H != d.Ho || d.y_W != d.Wo || d.out_off_y || d.out_off_x) return 0;
  if (d.Cin != 32 && d.Cin != 64 && d.Cin != 128 && d.Cin != 256) return 0;
  if (d.Cout % 32 || d.Cout > 1024 || d.w_cout_pad < d.Cout || d.x_pitch != d.Cin || d.y_pitch != d.Cout) return 0;
  const long M = (long)d.N * d.Ho * d.Wo;
  if (M < min_px || M * (d.Cin > d.Cout ? d.Cin : d.Cout) * 2 >= (1ll << 31)) return 0;
  if (d.bias || d.pro_mode || d.pro_side || d.pro_mask || d.pro_side2 || d.res_mask || (!d.scale) != (!d.shift)) return 0;
  if (d.res && d.res_pitch != d.Cout) return 0;
  if (((uintptr_t)d.x | (uintptr_t)d.y | (uintptr_t)d.w | (uintptr_t)d.res | (uintptr_t)d.bnb_x) % 16) return 0;
  if (d.bnb_x) {
    if (d.bnb_x_pitch != d.Cout || !d.stats || d.relu || d.scale) return 0;
    if (d.bnb_mask && (d.bnb_mask_pitch != d.Cout / 8 || (uintptr_t)d.bnb_mask % 2)) return 0;
    if (!d.bnb_consts && !(d.bnb_stats && d.bnb_gamma && d.bnb_beta)) return 0;
    if (!d.bnb_mask && !d.bnb_stats) return 0;       // a recomputed mask needs scale / shift: the bnb_stats form only
  }
  if (d.bnb_x && d.Cin == 256) return 0;           // 16 k-steps leave no registers for the prefetch next to the butterfly: the tile program is 10 % faster
  if (d.tail_mode) {         // the tail of a train-mode Bottleneck (conv_pw_tail_kernel): 32 / 64 -> a multiple of 64 channels
    if (d.tail_mode < 1 || d.tail_mode > 5 || (d.Cin != 32 && d.Cin != 64) || d.Cout % 64) return 0;
    if (d.tail_mode == 5 && (!d.tail_x2 || !d.tail_w2 || !d.tail_stats2 || !d.tail_gamma2 || !d.tail_beta2 || d.res ||
                             ((uintptr_t)d.tail_x2 | (uintptr_t)d.tail_w2) % 16)) return 0;
    if (d.bnb_x || d.scale || d.relu || d.stats == nullptr && (d.tail_mode == 1 || d.tail_mode == 3)) return 0;
    if (d.tail_mode != 1 && (!d.tail_stats || !d.tail_gamma || !d.tail_beta || !d.tail_mask || (uintptr_t)d.tail_mask % 2)) return 0;
    if (d.tail_mode == 2 && !d.res) return 0;
    if (d.tail_mode != 2 && d.res) return 0;
    if ((d.tail_mode == 3 || d.tail_mode == 4) && (!d.tail_g || (uintptr_t)d.tail_g % 16)) return 0;
    if (d.tail_mode == 4 && (!d.tail_bsums || (uintptr_t)d.tail_side % 16)) return 0;
  }
  p.ks = d.Cin / 16;
  p.mw = (p.ks <= 8 && d.Cout % 64 == 0) ? 2 : 1;
  const int nb = d.Cout / (32 * p.mw);
  p.wc = nb % 4 == 0 ? 4 : nb % 2 == 0 ? 2 : 1;
  p.wp = 4 / p.wc;
  p.groups = nb / p.wc;
  p.ntiles = (int)((M + 31) / 32);
  p.tpw = p.wgs = 0;       // filled by the launcher from the instantiation's occupancy (pw_fill_grid)
  return 1;
}

// LDS: [4 waves][2 blocks][64 lanes] statistic partials (waves that share channels are added up before the atomics), then rows
// 6 .. 9 of conv_row.h's constant table (epilogue reduce only)
constexpr int PW_STAT_BYTES = 4 * 2 * 64 * 4;
constexpr int PW_TAIL_XP_BYTES = 4 * 2 * 32 * 144;      // conv_pw_tail_kernel: two transpose tiles per wave (stores of modes 2 / 4)
// ... then one input tile per wave: 32 pixels x (Cin * 2 + 16) bytes (conv_pw_body_t: x arrives as whole rows)
static inline int pw_xtile_bytes(int Cin) { return 32 * (Cin * 2 + 16); }
static inline int pw_lds_bytes(const hrp_conv_desc& d) { return PW_STAT_BYTES + (d.bnb_x ? 10 * d.Cout * 4 : 0) + 4 * pw_xtile_bytes(d.Cin); }

template <int KS, int MW, bool EXT>
__device__ __forceinline__ void conv_pw_body_t(const hrp_conv_desc& d, const PwPlan& p, const int bid, const int stat_slot) {
  constexpr bool PF = KS <= 8 && !(KS == 8 && MW == 2);      // next tile's operand in flight under this tile's MFMAs (where the registers allow)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stat_lds = (float*)smem;
  float* ctab = (float*)(smem + PW_STAT_BYTES);      // rows 6 .. 9 of conv_row.h's constant table (epilogue reduce)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int Cin = KS * 16, Cout = d.Cout;
  const int grp = bid % p.groups, wg = bid / p.groups;
  const int wci = wave % p.wc, wpi = wave / p.wc;
  const int cbase = (grp * p.wc + wci) * (32 * MW);      // first output channel of the wave
  const bool bnb = d.bnb_x != nullptr;

  if (bnb) {
    for (int c = tid; c < Cout; c += 256) {
      if (d.bnb_stats) {
        float mean, inv, sc, sh;
        row_bn_consts(d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps, c, Cout, mean, inv, sc, sh);
        ctab[8 * Cout + c] = sc; ctab[9 * Cout + c] = sh;
        ctab[6 * Cout + c] = inv; ctab[7 * Cout + c] = -mean * inv;
      } else {
        const float mean = d.bnb_consts[c], inv = d.bnb_consts[Cout + c];
        ctab[8 * Cout + c] = 0.f; ctab[9 * Cout + c] = 0.f;
        ctab[6 * Cout + c] = inv; ctab[7 * Cout + c] = -mean * inv;
      }
    }
    __syncthreads();
  }

  // ---- weights: MFMA row rho = l31 carries output channel 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3) of the block
  bf16x8 wf[MW][KS];
  {
    const int co_lane = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    const char* wl = (const char*)d.w + (size_t)(d.wtap[0] * d.w_cout_pad + cbase + co_lane) * ROW + half * 16;
    const size_t kstride = (size_t)d.w_ntaps * d.w_cout_pad * ROW;
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) wf[mi][kk] = *(const bf16x8*)(wl + kk * kstride + mi * 32 * ROW);
  }

  // ---- the wave's tiles: t0, t0 + wp, ...   (the workgroup's run is contiguous in memory)
  const long M = (long)d.N * d.Ho * d.Wo;
  const int tile0 = wg * (p.tpw * p.wp) + wpi, tstep = p.wp;
  // x as whole rows (Cin * 2 bytes = 2 KS pieces of 16 bytes per pixel): lane -> (pixel lane / (2 KS) + (32 / KS) k, piece lane % (2 KS)), KS
  // vectors per lane and tile, re-read as MFMA B fragments (pixel l31, k half) from the wave's LDS tile.  In MFMA order a load
  // instruction touched 32 lines for 1 KB (two pieces of every pixel's row); at 512-byte rows (Cin = 256) the pass ran at 3.3 TB/s.
  constexpr int XPITCH = KS * 32 + 16;
  char* xt = smem + PW_STAT_BYTES + (bnb ? 10 * Cout * 4 : 0) + wave * (32 * XPITCH);
  const int xr_px = lane / (2 * KS), xr_pc = (lane % (2 * KS)) * 16;
  auto load_tile = [&](int t, uint4 (&raw)[KS]) {
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const long pix = (long)t * 32 + xr_px + (32 / KS) * k;
      raw[k] = (t < p.ntiles && pix < M) ? *(const uint4*)((const char*)d.x + (size_t)pix * (Cin * 2) + xr_pc) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stage_x = [&](const uint4 (&raw)[KS], bf16x8 (&xb)[KS]) {
#pragma unroll
    for (int k = 0; k < KS; ++k) *(uint4*)(xt + (xr_px + (32 / KS) * k) * XPITCH + xr_pc) = raw[k];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) xb[kk] = *(const bf16x8*)(xt + l31 * XPITCH + kk * 32 + half * 16);
  };

  // Statistics.  Plain sum / sum of squares (train-mode forward layers): a SECOND product per tile with the operands swapped -
  // the x fragment is as well the A operand of pixel rows, the weight fragment the B operand of channel columns - gives the
  // transposed tile, lane = one channel (co_lane of l31) x 16 pixels: its sums are 32 VALU operations on the lane's own
  // registers, accumulated over all tiles of the workgroup (the values rounded to bf16 as the stored ones are).  KS more MFMAs
  // per tile on an idle matrix pipe instead of a 31-shuffle butterfly per tile (measured 64 -> 256: 50 -> 38 us).
  // The BatchNorm-backward sums need the BatchNorm input and the mask at the same positions: those launches fold every tile's
  // sums with the reduce-scatter butterfly of conv_row.h (lane l31 < 16 keeps sum 1 of channel cl + l31, else sum 2).
  const bool tstats = d.stats && !bnb && !d.scale && !d.res && !d.relu;
  float vt[MW], vq[MW];
#pragma unroll
  for (int mi = 0; mi < MW; ++mi) vt[mi] = vq[mi] = 0.f;

  // Whole-row epilogue (ROWIO: two channel blocks per wave = 128 bytes of every pixel's row, no epilogue reduce, no extended
  // options): the residual arrives as rows one tile ahead with x, goes through the wave's LDS tile into MFMA order, the results go
  // back through the tile and leave as 128-byte rows (as conv_pw_tail_kernel; in MFMA order a store instruction touched 32 lines).
  constexpr bool ROWIO_T = !EXT && MW == 2 && KS >= 4 && KS <= 8;
  const bool rowio = ROWIO_T && !bnb && !d.stats;      // (with the transposed-product statistics the direct stores are the faster form: 32.6 against 37.2 us)
  const int xp_mine = l31 * XPITCH + half * 32, xp_row = lane >> 3, xp_pc = (lane & 7) * 16;
  auto load_res = [&](int t, uint4 (&rr)[4]) {
    if (ROWIO_T && rowio && d.res) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long px = (long)t * 32 + xp_row + 8 * k;
        rr[k] = (t < p.ntiles && px < M) ? *(const uint4*)((const char*)d.res + (size_t)px * (Cout * 2) + cbase * 2 + xp_pc) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto compute = [&](int t, const uint4 (&xraw)[KS], const uint4 (&rraw)[4]) {
    const long pix = (long)t * 32 + l31;
    const unsigned ok = (t < p.ntiles && pix < M) ? 1u : 0u;
    bf16x8 xb[KS];
    stage_x(xraw, xb);
    if (ROWIO_T && rowio) {
      f32x16 acc2[MW];
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[mi][i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc2[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[mi][kk], xb[kk], acc2[mi], 0, 0, 0);
      }
      if (d.res) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(uint4*)(xt + (xp_row + 8 * k) * XPITCH + xp_pc) = rraw[k];
      }
      const bool aff = d.scale != nullptr;
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) {
        const int cl = cbase + mi * 32 + 16 * half;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = acc2[mi][8 * hh + i];
          if (aff) {
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
              const float4 a4 = *(const float4*)(d.scale + cl + 8 * hh + i), b4 = *(const float4*)(d.shift + cl + 8 * hh + i);
              v[i] = v[i] * a4.x + b4.x; v[i + 1] = v[i + 1] * a4.y + b4.y; v[i + 2] = v[i + 2] * a4.z + b4.z; v[i + 3] = v[i + 3] * a4.w + b4.w;
            }
          }
          char* q = xt + xp_mine + mi * 64 + hh * 16;
          if (d.res) {
            float r[8];
            Elem<bf16_t>::unpack(*(const uint4*)q, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += r[i];
          }
          if (d.relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
          }
          *(uint4*)q = Elem<bf16_t>::pack(v);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long px = (long)t * 32 + xp_row + 8 * k;
        if (t < p.ntiles && px < M)
          *(uint4*)((char*)d.y + (size_t)px * (Cout * 2) + cbase * 2 + xp_pc) = *(const uint4*)(xt + (xp_row + 8 * k) * XPITCH + xp_pc);
      }
      return;
    }
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
      f32x16 acc[1];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[mi][kk], xb[kk], acc[0], 0, 0, 0);
      if (tstats) {
        f32x16 tr;
#pragma unroll
        for (int i = 0; i < 16; ++i) tr[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[kk], wf[mi][kk], tr, 0, 0, 0);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = tr[8 * hh + i];
          Elem<bf16_t>::unpack(Elem<bf16_t>::pack(v), v);          // the values as stored (pixels outside the tensor: zero)
#pragma unroll
          for (int i = 0; i < 8; ++i) { vt[mi] += v[i]; vq[mi] = fmaf(v[i], v[i], vq[mi]); }
        }
      }
      float s1[16], s2[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
      const int cl = cbase + mi * 32 + 16 * half;
      const unsigned off[1] = {(unsigned)(pix * Cout + cl) * 2u};
      if (tstats) {
        row_epilogue<1, EXT, false>(d, acc, off, ok, cl, ctab, Cout, false, s1, s2);
      } else {
        row_epilogue<1, EXT>(d, acc, off, ok, cl, ctab, Cout, bnb, s1, s2);
        if (d.stats) vt[mi] += row_reduce32(s1, s2, l31);
      }
    }
  };

  if constexpr (PF) {
    uint4 xa[KS], xb[KS], ra[4], rb[4];
    load_tile(tile0, xa);
    load_res(tile0, ra);
    for (int it = 0; it < p.tpw; it += 2) {
      load_tile(tile0 + (it + 1) * tstep, xb);
      load_res(tile0 + (it + 1) * tstep, rb);
      compute(tile0 + it * tstep, xa, ra);
      if (it + 2 < p.tpw) { load_tile(tile0 + (it + 2) * tstep, xa); load_res(tile0 + (it + 2) * tstep, ra); }
      if (it + 1 < p.tpw) compute(tile0 + (it + 1) * tstep, xb, rb);
    }
  } else {
    for (int it = 0; it < p.tpw; ++it) {
      uint4 xa[KS], ra[4];
      load_tile(tile0 + it * tstep, xa);
      load_res(tile0 + it * tstep, ra);
      compute(tile0 + it * tstep, xa, ra);
    }
  }

  if (d.stats) {
    // per lane ONE value per channel block.  tstats: half 0 = sum, half 1 = sum of squares of channel co_lane(l31);
    // butterfly: l31 < 16 sum 1, l31 >= 16 sum 2 of channel cl + (l31 & 15).  Waves over different pixel tiles hold the same
    // channels in the same lanes: added up through LDS, then one atomic per channel and workgroup.
    float v[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
      v[mi] = vt[mi];
      if (tstats) {
        const float a = vt[mi] + __shfl_xor(vt[mi], 32, 64), b = vq[mi] + __shfl_xor(vq[mi], 32, 64);      // 16 + 16 pixel rows
        v[mi] = half ? b : a;
      }
    }
    if (p.wp > 1) {
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) stat_lds[(wave * 2 + mi) * 64 + lane] = v[mi];
      __syncthreads();
      if (wpi != 0) return;
#pragma unroll
      for (int mi = 0; mi < MW; ++mi)
        for (int q = 1; q < p.wp; ++q) v[mi] += stat_lds[((wci + q * p.wc) * 2 + mi) * 64 + lane];
    }
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
      if (tstats) {
        const int co_lane = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
        atomicAdd(d.stats + stat_slot * 2 * Cout + half * Cout + cbase + mi * 32 + co_lane, (double)v[mi]);
      } else {
        row_stats_commit(d, v[mi], l31, cbase + mi * 32 + 16 * half, ctab, Cout, bnb, stat_slot);
      }
    }
  }
}

template <int KS, int MW>
__device__ __forceinline__ void conv_pw_body(const hrp_conv_desc& d, const PwPlan& p, const int bid, const int stat_slot) {
  if (row_ext(d)) conv_pw_body_t<KS, MW, true>(d, p, bid, stat_slot);
  else conv_pw_body_t<KS, MW, false>(d, p, bid, stat_slot);
}

template <int KS, int MW>
__global__ __launch_bounds__(256, 2) void conv_pw_kernel(const hrp_conv_desc d, const PwPlan p) {
  conv_pw_body<KS, MW>(d, p, blockIdx.x, blockIdx.x & (HRP_STAT_SLOTS - 1));
}

// ---- the tail of a train-mode Bottleneck: out = relu(bn3(conv3(h)) + shortcut) (reference HRnet.py:88-96) and its backward without
// the raw product in HBM (hrp_conv_desc.tail_mode, include/hrp.h).  The product of a 64 -> 256 layer costs 33 MB of reads; its
// output, stored and read back, 2 x 134 MB - so every pass that needs it multiplies again:
//   MODE 1  statistics (sum, sum of squares of the unrounded product) from the TRANSPOSED product alone - lane = one channel x 16 pixels
//   MODE 2  y = relu(acc * sc + sh + res), ReLU bits to tail_mask                                (replaces the store of y3 + hrp_ew_fwd)
//   MODE 3  s1 += g, s2 += g * acc with g = tail_g under the bits; one butterfly per workgroup   (replaces hrp_ew_bwd_reduce)
//   MODE 4  y = sc * g + (c1 * acc + c0), tail_side (+)= g                                       (replaces hrp_ew_bwd_apply)
// A lane's 16 channels are the same in every tile: the per-channel constants live in registers.
template <int KS, int MODE>
__global__ __launch_bounds__(256, 2) void conv_pw_tail_kernel(const hrp_conv_desc d, const PwPlan p) {
  constexpr int MW = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stat_lds = (float*)smem;
  const int bid = blockIdx.x, stat_slot = blockIdx.x & (HRP_STAT_SLOTS - 1);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int Cin = KS * 16, Cout = d.Cout;
  const int grp = bid % p.groups, wg = bid / p.groups;
  const int wci = wave % p.wc, wpi = wave / p.wc;
  const int cbase = (grp * p.wc + wci) * (32 * MW);
  const int co_lane = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);

  bf16x8 wf[MW][KS];
  {
    const char* wl = (const char*)d.w + (size_t)(d.wtap[0] * d.w_cout_pad + cbase + co_lane) * ROW + half * 16;
    const size_t kstride = (size_t)d.w_ntaps * d.w_cout_pad * ROW;
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) wf[mi][kk] = *(const bf16x8*)(wl + kk * kstride + mi * 32 * ROW);
  }

  // per-channel constants: a table in LDS (one thread per channel derives them from the statistic slots), then the lane's
  // channels cl(mi) .. + 15 into registers - a lane's channels are the same in every tile
  float* ctab = (float*)(smem + PW_STAT_BYTES);          // [3][Cout]: sc, sh | sc, c0, c1
  // Stores through a transpose (MODE 2 / 4): a lane holds 2 x 32 bytes of ONE pixel's row, so a store instruction of the direct form
  // scatters 64 pieces of 16 bytes over 64 lines and relies on L2 to merge eight such pieces per 128-byte line before it evicts
  // them - it does at 128 output channels (5.5 TB/s) and does not at 256 (3.6).  Each wave drops its 32 pixels x 128 bytes into a
  // private LDS tile ([pixel][144 B]) and stores them back out as whole 128-byte rows, eight lanes per row.
  constexpr int XP_PITCH = 144, XP_BYTES = 32 * XP_PITCH;
  char* xp_y = smem + PW_STAT_BYTES + 3 * Cout * 4 + wave * (2 * XP_BYTES);
  char* xp_s = xp_y + XP_BYTES;                              // (MODE 4: the rider)
  const int xp_mine = l31 * XP_PITCH + half * 32;          // + mi * 64 + hh * 16: the lane's pieces inside its pixel's 128 bytes
  const int xp_row = lane >> 3, xp_pc = (lane & 7) * 16;   // read-back: pixel xp_row + 8 k, piece lane & 7
  float k_sc[MODE == 2 || MODE == 4 ? MW : 1][16], k_sh[MODE == 2 ? MW : 1][16];
  if constexpr (MODE == 2 || MODE == 4) {
    for (int c = tid; c < Cout; c += 256) {
      float mean, inv, sc, sh;
      row_bn_consts(d.tail_stats, d.tail_gamma, d.tail_beta, d.tail_count, d.tail_eps, c, Cout, mean, inv, sc, sh);
      ctab[c] = sc;
      if constexpr (MODE == 2) ctab[Cout + c] = sh;
      else {         // sc * (g - k0 - xhat * k1), xhat = inv * y - mean * inv  ==  sc * g + (c1 * y + c0)
        const float k0 = slot_sum(d.tail_bsums, c, 2 * Cout) / d.tail_count, k1 = slot_sum(d.tail_bsums, Cout + c, 2 * Cout) / d.tail_count;
        ctab[Cout + c] = -sc * fmaf(k1, -mean * inv, k0);
        ctab[2 * Cout + c] = -sc * k1 * inv;
      }
    }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
      const int cl = cbase + mi * 32 + 16 * half;
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        const float4 a4 = *(const float4*)(ctab + cl + i), b4 = *(const float4*)(ctab + Cout + cl + i);
        k_sc[mi][i] = a4.x; k_sc[mi][i + 1] = a4.y; k_sc[mi][i + 2] = a4.z; k_sc[mi][i + 3] = a4.w;
        if constexpr (MODE == 2) { k_sh[mi][i] = b4.x; k_sh[mi][i + 1] = b4.y; k_sh[mi][i + 2] = b4.z; k_sh[mi][i + 3] = b4.w; }
      }
    }
  }

  const long M = (long)d.N * d.Ho * d.Wo;
  const int tile0 = wg * (p.tpw * p.wp) + wpi, tstep = p.wp;
  // x as whole rows (Cin * 2 bytes = 2 KS pieces per pixel): lane -> (pixel lane / (2 KS) + (32 / KS) k, piece lane % (2 KS)), KS vectors
  // per lane and tile - the same registers as the MFMA-ordered form (pixel l31, half) needed, but a load instruction touches 8 lines
  // instead of 32.  stage_x puts a tile's rows into the wave's LDS tile and reads the B fragments back.
  using XRaw = uint4[KS];
  const int xr_px = lane / (2 * KS), xr_pc = (lane % (2 * KS)) * 16;
  auto load_tile = [&](int t, uint4 (&raw)[KS]) {
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const long pix = (long)t * 32 + xr_px + (32 / KS) * k;
      raw[k] = (t < p.ntiles && pix < M) ? *(const uint4*)((const char*)d.x + (size_t)pix * (Cin * 2) + xr_pc) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stage_x = [&](const uint4 (&raw)[KS], bf16x8 (&xb)[KS]) {
#pragma unroll
    for (int k = 0; k < KS; ++k) *(uint4*)(xp_y + (xr_px + (32 / KS) * k) * XP_PITCH + xr_pc) = raw[k];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) xb[kk] = *(const bf16x8*)(xp_y + l31 * XP_PITCH + kk * 32 + half * 16);
  };

  float vt[MW], vq[MW];                      // MODE 1: per lane (channel co_lane) sum / sum of squares
  float s1[MODE == 3 ? MW : 1][16], s2[MODE == 3 ? MW : 1][16];
#pragma unroll
  for (int mi = 0; mi < MW; ++mi) vt[mi] = vq[mi] = 0.f;
  if constexpr (MODE == 3) {
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
      for (int i = 0; i < 16; ++i) s1[mi][i] = s2[mi][i] = 0.f;
  }

  // operands of the epilogue at the lane's 2 x 16 channels of one pixel: MODE 2 the shortcut, MODE 3 / 4 the gradient of the block
  // output and its ReLU bits (+ the rider's old value when it accumulates).  Loaded one tile AHEAD, with x: issued inside the tile
  // that uses them they were one exposed HBM latency per tile (mode 2: 99 us for 310 MB; the element-wise pass it replaces streams
  // the same tensors at 4.6 TB/s)
  struct Epi {
    uint4 raw[4];                       // the wave's 32 pixels x 128 bytes as whole rows: pixel xp_row + 8 k, piece lane & 7
    uint4 sraw[MODE == 4 ? 4 : 1];
    int bits[MW];
  };
  const bool side_acc = MODE == 4 && d.tail_side && d.tail_side_acc;
  auto load_epi = [&](int t, Epi& e) {
    if constexpr (MODE != 1) {
      const char* src = (const char*)(MODE == 2 ? d.res : d.tail_g);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long px = (long)t * 32 + xp_row + 8 * k;
        const size_t off = (size_t)px * (Cout * 2) + cbase * 2 + xp_pc;
        e.raw[k] = make_uint4(0, 0, 0, 0);
        if (t < p.ntiles && px < M) {
          e.raw[k] = *(const uint4*)(src + off);
          if constexpr (MODE == 4) {
            if (side_acc) e.sraw[k] = *(const uint4*)((const char*)d.tail_side + off);
          }
        }
      }
      if constexpr (MODE >= 3) {
        const long pix = (long)t * 32 + l31;
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
          e.bits[mi] = 0;
          if (t < p.ntiles && pix < M) e.bits[mi] = *(const unsigned short*)(d.tail_mask + (((size_t)pix * Cout + cbase + mi * 32 + 16 * half) >> 3));
        }
      }
    }
  };
  // ... and back into the MFMA's lane order (pixel l31, 16 channels per half) through the same LDS tiles the stores use: the rows go
  // in here, every lane reads ITS pieces where it needs them (and later writes its results to the same places)
  auto stage_epi = [&](const uint4 (&raw)[4], char* xp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) *(uint4*)(xp + (xp_row + 8 * k) * XP_PITCH + xp_pc) = raw[k];
  };

  auto flush = [&](const char* xp, void* dst, int t) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long px = (long)t * 32 + xp_row + 8 * k;
      if (t < p.ntiles && px < M)
        *(uint4*)((char*)dst + (size_t)px * (Cout * 2) + cbase * 2 + xp_pc) = *(const uint4*)(xp + (xp_row + 8 * k) * XP_PITCH + xp_pc);
    }
  };
  auto compute = [&](int t, const uint4 (&xraw)[KS], const Epi& e) {
    const long pix = (long)t * 32 + l31;
    const bool ok = t < p.ntiles && pix < M;
    bf16x8 xb[KS];
    stage_x(xraw, xb);
    if constexpr (MODE != 1) stage_epi(e.raw, xp_y);
    if constexpr (MODE == 4) {
      if (side_acc) stage_epi(e.sraw, xp_s);
    }
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
      if constexpr (MODE == 1) {
        f32x16 tr;
#pragma unroll
        for (int i = 0; i < 16; ++i) tr[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) tr = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[kk], wf[mi][kk], tr, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) { vt[mi] += tr[i]; vq[mi] = fmaf(tr[i], tr[i], vq[mi]); }      // (pixels outside the tensor: zero rows)
      } else {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[mi][kk], xb[kk], acc, 0, 0, 0);
        const int cl = cbase + mi * 32 + 16 * half;
        const unsigned off = (unsigned)(pix * Cout + cl) * 2u;
        if constexpr (MODE == 2) {
          unsigned bits = 0;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            float v[8], r[8];
            Elem<bf16_t>::unpack(*(const uint4*)(xp_y + xp_mine + mi * 64 + hh * 16), r);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              v[i] = fmaxf(fmaf(acc[8 * hh + i], k_sc[mi][8 * hh + i], k_sh[mi][8 * hh + i]) + r[i], 0.f);
              bits |= (v[i] > 0.f ? 1u : 0u) << (8 * hh + i);
            }
            *(uint4*)(xp_y + xp_mine + mi * 64 + hh * 16) = Elem<bf16_t>::pack(v);
          }
          if (ok) *(unsigned short*)(d.tail_mask + (off >> 4)) = (unsigned short)bits;
        } else {
          const int bits = e.bits[mi];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            float g[8];
            Elem<bf16_t>::unpack(*(const uint4*)(xp_y + xp_mine + mi * 64 + hh * 16), g);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = row_keep_if_bit(g[i], bits, 8 * hh + i);
            if constexpr (MODE == 3) {
#pragma unroll
              for (int i = 0; i < 8; ++i) { s1[mi][8 * hh + i] += g[i]; s2[mi][8 * hh + i] = fmaf(g[i], acc[8 * hh + i], s2[mi][8 * hh + i]); }
            } else {
              if (d.tail_side) {
                char* q = xp_s + xp_mine + mi * 64 + hh * 16;
                if (side_acc) {
                  float o[8];
                  Elem<bf16_t>::unpack(*(const uint4*)q, o);
#pragma unroll
                  for (int i = 0; i < 8; ++i) o[i] += g[i];
                  *(uint4*)q = Elem<bf16_t>::pack(o);
                } else {
                  *(uint4*)q = Elem<bf16_t>::pack(g);
                }
              }
              float v[8];
#pragma unroll
              for (int i = 0; i < 8; i += 4) {          // c0 / c1 from the LDS table per tile (as registers they spill next to the prefetched operands)
                const float4 c0 = *(const float4*)(ctab + Cout + cl + 8 * hh + i), c1 = *(const float4*)(ctab + 2 * Cout + cl + 8 * hh + i);
                v[i] = fmaf(k_sc[mi][8 * hh + i], g[i], fmaf(c1.x, acc[8 * hh + i], c0.x));
                v[i + 1] = fmaf(k_sc[mi][8 * hh + i + 1], g[i + 1], fmaf(c1.y, acc[8 * hh + i + 1], c0.y));
                v[i + 2] = fmaf(k_sc[mi][8 * hh + i + 2], g[i + 2], fmaf(c1.z, acc[8 * hh + i + 2], c0.z));
                v[i + 3] = fmaf(k_sc[mi][8 * hh + i + 3], g[i + 3], fmaf(c1.w, acc[8 * hh + i + 3], c0.w));
              }
              *(uint4*)(xp_y + xp_mine + mi * 64 + hh * 16) = Elem<bf16_t>::pack(v);
            }
          }
        }
      }
    }
    if constexpr (MODE == 2 || MODE == 4) {
      flush(xp_y, d.y, t);
      if constexpr (MODE == 4) {
        if (d.tail_side) flush(xp_s, d.tail_side, t);
      }
    }
  };

  if constexpr (MODE == 1) {
    // statistics only: nothing but x is read - four tiles of loads in flight per wave (two left the pass latency bound: 24 us for
    // 33 MB at 64 -> 256 channels)
    uint4 xq[4][KS];
    Epi e0;
#pragma unroll
    for (int j = 0; j < 4; ++j) load_tile(tile0 + j * tstep, xq[j]);
    for (int it = 0; it < p.tpw; it += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (it + j < p.tpw) compute(tile0 + (it + j) * tstep, xq[j], e0);
        if (it + j + 4 < p.tpw) load_tile(tile0 + (it + j + 4) * tstep, xq[j]);
      }
    }
  } else {
    uint4 xa[KS], xb[KS];
    Epi ea, eb;
    load_tile(tile0, xa);
    load_epi(tile0, ea);
    for (int it = 0; it < p.tpw; it += 2) {
      load_tile(tile0 + (it + 1) * tstep, xb);
      load_epi(tile0 + (it + 1) * tstep, eb);
      compute(tile0 + it * tstep, xa, ea);
      if (it + 2 < p.tpw) { load_tile(tile0 + (it + 2) * tstep, xa); load_epi(tile0 + (it + 2) * tstep, ea); }
      if (it + 1 < p.tpw) compute(tile0 + (it + 1) * tstep, xb, eb);
    }
  }

  if constexpr (MODE == 1 || MODE == 3) {
    // one value per lane and channel block.  MODE 1: half 0 = sum, half 1 = sum of squares of channel co_lane.  MODE 3 (after the
    // reduce-scatter butterfly): l31 < 16 the sum of g, l31 >= 16 the sum of g * y of channel cl + (l31 & 15).
    float v[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
      if constexpr (MODE == 1) {
        const float a = vt[mi] + __shfl_xor(vt[mi], 32, 64), b = vq[mi] + __shfl_xor(vq[mi], 32, 64);
        v[mi] = half ? b : a;
      } else {
        v[mi] = row_reduce32(s1[mi], s2[mi], l31);
      }
    }
    if (p.wp > 1) {
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) stat_lds[(wave * 2 + mi) * 64 + lane] = v[mi];
      __syncthreads();
      if (wpi != 0) return;
#pragma unroll
      for (int mi = 0; mi < MW; ++mi)
        for (int q = 1; q < p.wp; ++q) v[mi] += stat_lds[((wci + q * p.wc) * 2 + mi) * 64 + lane];
    }
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
      if constexpr (MODE == 1) {
        atomicAdd(d.stats + stat_slot * 2 * Cout + half * Cout + cbase + mi * 32 + co_lane, (double)v[mi]);
      } else {
        const int which = l31 >> 4, c = cbase + mi * 32 + 16 * half + (l31 & 15);
        const float other = __shfl_xor(v[mi], 16, 64);          // sum g of the same channel, for the lanes holding sum g * y
        float tot = v[mi];
        if (which == 1) {                                       // sum g * xhat = inv * sum g y - mean * inv * sum g
          float mean, inv, sc, sh;
          row_bn_consts(d.tail_stats, d.tail_gamma, d.tail_beta, d.tail_count, d.tail_eps, c, Cout, mean, inv, sc, sh);
          tot = fmaf(inv, v[mi], -mean * inv * other);
        }
        atomicAdd(d.stats + stat_slot * 2 * Cout + which * Cout + c, (double)tot);
      }
    }
  }
}

// MODE 5: the projection form, out = relu(bn(w h) + bn2(w2 x2)): two products per tile, the constants of both BatchNorms read
// from the LDS table per tile (two weight sets and two operand prefetches leave no registers for 128 constants).
template <int KS>
__global__ __launch_bounds__(256, 2) void conv_pw_tail2_kernel(const hrp_conv_desc d, const PwPlan p) {
  constexpr int MW = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ctab = (float*)(smem + PW_STAT_BYTES);          // [4][Cout]: sc, sh, sc2, sh2 (sh holds sh + sh2, row 3 unused)
  const int bid = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int Cin = KS * 16, Cout = d.Cout;
  const int grp = bid % p.groups, wg = bid / p.groups;
  const int wci = wave % p.wc, wpi = wave / p.wc;
  const int cbase = (grp * p.wc + wci) * (32 * MW);
  const int co_lane = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  for (int c = tid; c < Cout; c += 256) {
    float mean, inv, sc, sh, sc2, sh2;
    row_bn_consts(d.tail_stats, d.tail_gamma, d.tail_beta, d.tail_count, d.tail_eps, c, Cout, mean, inv, sc, sh);
    row_bn_consts(d.tail_stats2, d.tail_gamma2, d.tail_beta2, d.tail_count, d.tail_eps, c, Cout, mean, inv, sc2, sh2);
    ctab[c] = sc; ctab[Cout + c] = sh + sh2; ctab[2 * Cout + c] = sc2;
  }
  bf16x8 wf[MW][KS], wg2[MW][KS];
  {
    const size_t lane_off = (size_t)(d.wtap[0] * d.w_cout_pad + cbase + co_lane) * ROW + half * 16;
    const size_t kstride = (size_t)d.w_ntaps * d.w_cout_pad * ROW;
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        wf[mi][kk] = *(const bf16x8*)((const char*)d.w + lane_off + kk * kstride + mi * 32 * ROW);
        wg2[mi][kk] = *(const bf16x8*)((const char*)d.tail_w2 + lane_off + kk * kstride + mi * 32 * ROW);
      }
  }
  __syncthreads();
  const long M = (long)d.N * d.Ho * d.Wo;
  const int tile0 = wg * (p.tpw * p.wp) + wpi, tstep = p.wp;
  // stores as whole 128-byte rows through a per-wave LDS tile (as conv_pw_tail_kernel)
  constexpr int XP_PITCH = 144, XP_BYTES = 32 * XP_PITCH;
  char* xp_y = smem + PW_STAT_BYTES + 3 * Cout * 4 + wave * XP_BYTES;
  const int xp_mine = l31 * XP_PITCH + half * 32, xp_row = lane >> 3, xp_pc = (lane & 7) * 16;
  auto load_tile = [&](int t, bf16x8 (&xb)[KS], bf16x8 (&xc)[KS]) {
    const long pix = (long)t * 32 + l31;
    if (t < p.ntiles && pix < M) {
      const char* q = (const char*)d.x + half * 16 + (size_t)pix * (Cin * 2);
      const char* q2 = (const char*)d.tail_x2 + half * 16 + (size_t)pix * (Cin * 2);
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) { xb[kk] = *(const bf16x8*)(q + kk * 32); xc[kk] = *(const bf16x8*)(q2 + kk * 32); }
    } else {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int i = 0; i < 8; ++i) { xb[kk][i] = (__bf16)0.f; xc[kk][i] = (__bf16)0.f; }
    }
  };
  auto compute = [&](int t, const bf16x8 (&xb)[KS], const bf16x8 (&xc)[KS]) {
    const long pix = (long)t * 32 + l31;
    const bool ok = t < p.ntiles && pix < M;
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
      f32x16 acc, acc2;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = acc2[i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[mi][kk], xb[kk], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wg2[mi][kk], xc[kk], acc2, 0, 0, 0);
      }
      const int cl = cbase + mi * 32 + 16 * half;
      const unsigned off = (unsigned)(pix * Cout + cl) * 2u;
      unsigned bits = 0;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i += 4) {
          const float4 a4 = *(const float4*)(ctab + cl + 8 * hh + i), b4 = *(const float4*)(ctab + Cout + cl + 8 * hh + i);
          const float4 c4 = *(const float4*)(ctab + 2 * Cout + cl + 8 * hh + i);
          const float sa[4] = {a4.x, a4.y, a4.z, a4.w}, sb[4] = {b4.x, b4.y, b4.z, b4.w}, sc2[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[i + e] = fmaxf(fmaf(acc[8 * hh + i + e], sa[e], fmaf(acc2[8 * hh + i + e], sc2[e], sb[e])), 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) bits |= (v[i] > 0.f ? 1u : 0u) << (8 * hh + i);
        *(uint4*)(xp_y + xp_mine + mi * 64 + hh * 16) = Elem<bf16_t>::pack(v);
      }
      if (ok) *(unsigned short*)(d.tail_mask + (off >> 4)) = (unsigned short)bits;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long px = (long)t * 32 + xp_row + 8 * k;
      if (t < p.ntiles && px < M)
        *(uint4*)((char*)d.y + (size_t)px * (Cout * 2) + cbase * 2 + xp_pc) = *(const uint4*)(xp_y + (xp_row + 8 * k) * XP_PITCH + xp_pc);
    }
  };
  bf16x8 xa[KS], xa2[KS], xb[KS], xb2[KS];
  load_tile(tile0, xa, xa2);
  for (int it = 0; it < p.tpw; it += 2) {
    load_tile(tile0 + (it + 1) * tstep, xb, xb2);
    compute(tile0 + it * tstep, xa, xa2);
    if (it + 2 < p.tpw) load_tile(tile0 + (it + 2) * tstep, xa, xa2);
    if (it + 1 < p.tpw) compute(tile0 + (it + 1) * tstep, xb, xb2);
  }
}

template <int KS, int MODE>
static inline int pw_tail_occupancy() {
  static int occ = 0;
  if (!occ) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_pw_tail_kernel<KS, MODE>, 256, PW_STAT_BYTES + 3 * 256 * 4 + PW_TAIL_XP_BYTES) != hipSuccess || nb < 1) nb = 1;
    occ = nb;
  }
  return occ;
}

// ONE round of workgroups: as many as are resident at once (all of them run side by side and finish together - a second,
// partly filled round would idle most of the chip), each walking ntiles / (that many) tiles
static inline void pw_fill_grid(PwPlan& p, int occupancy) {
  const int cus = 256;
  long per_group = (long)cus * occupancy / p.groups;
  if (per_group < 1) per_group = 1;
  long tpw = (p.ntiles + per_group * p.wp - 1) / (per_group * p.wp);
  if (tpw < 2) tpw = 2;
  p.tpw = (int)tpw;
  p.wgs = (p.ntiles + p.tpw * p.wp - 1) / (p.tpw * p.wp);
}

template <int KS, int MW>
static inline int pw_occupancy() {
  static int occ = 0;
  if (!occ) {
    int nb = 0;
    (void)hipFuncSetAttribute((const void*)conv_pw_kernel<KS, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_pw_kernel<KS, MW>, 256, PW_STAT_BYTES + 4 * pw_xtile_bytes(KS * 16)) != hipSuccess || nb < 1) nb = 2;
    occ = nb;
  }
  return occ;
}

static inline int launch_conv_pw(const hrp_conv_desc& d, const PwPlan& p0, hipStream_t s) {
  PwPlan p = p0;
  const dim3 blk(256);
  const int lds = pw_lds_bytes(d);
  if (d.tail_mode) {
#define HRP_PWT_CASE(K, MD) if (p.ks == K && d.tail_mode == MD) { pw_fill_grid(p, pw_tail_occupancy<K, MD>()); hipLaunchKernelGGL((conv_pw_tail_kernel<K, MD>), dim3(p.wgs * p.groups), blk, PW_STAT_BYTES + 3 * d.Cout * 4 + PW_TAIL_XP_BYTES, s, d, p); return check_launch("conv_pw_tail_kernel"); }
    HRP_PWT_CASE(2, 1) HRP_PWT_CASE(2, 2) HRP_PWT_CASE(2, 3) HRP_PWT_CASE(2, 4) HRP_PWT_CASE(4, 1) HRP_PWT_CASE(4, 2) HRP_PWT_CASE(4, 3) HRP_PWT_CASE(4, 4)
#undef HRP_PWT_CASE
    if (d.tail_mode == 5) {
      static int occ2[2] = {0, 0};
      const int which = p.ks == 2 ? 0 : 1;
      if (!occ2[which]) {
        int nb = 0;
        const hipError_t e = p.ks == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_pw_tail2_kernel<2>, 256, PW_STAT_BYTES + 3 * 256 * 4 + PW_TAIL_XP_BYTES)
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_pw_tail2_kernel<4>, 256, PW_STAT_BYTES + 3 * 256 * 4 + PW_TAIL_XP_BYTES);
        occ2[which] = (e != hipSuccess || nb < 1) ? 1 : nb;
      }
      pw_fill_grid(p, occ2[which]);
      if (p.ks == 2) hipLaunchKernelGGL((conv_pw_tail2_kernel<2>), dim3(p.wgs * p.groups), blk, PW_STAT_BYTES + 3 * d.Cout * 4 + PW_TAIL_XP_BYTES, s, d, p);
      else hipLaunchKernelGGL((conv_pw_tail2_kernel<4>), dim3(p.wgs * p.groups), blk, PW_STAT_BYTES + 3 * d.Cout * 4 + PW_TAIL_XP_BYTES, s, d, p);
      return check_launch("conv_pw_tail2_kernel");
    }
    set_error("conv: no Bottleneck-tail instantiation for Cin=%d mode %d", d.Cin, d.tail_mode);
    return HRP_ERR_ARG;
  }
#define HRP_PW_CASE(K, M_) if (p.ks == K && p.mw == M_) { pw_fill_grid(p, pw_occupancy<K, M_>()); hipLaunchKernelGGL((conv_pw_kernel<K, M_>), dim3(p.wgs * p.groups), blk, lds, s, d, p); return check_launch("conv_pw_kernel"); }
  HRP_PW_CASE(2, 1) HRP_PW_CASE(2, 2) HRP_PW_CASE(4, 1) HRP_PW_CASE(4, 2) HRP_PW_CASE(8, 1) HRP_PW_CASE(8, 2) HRP_PW_CASE(16, 1)
#undef HRP_PW_CASE
  set_error("conv: no pointwise instantiation for Cin=%d (ks %d, mw %d)", d.Cin, p.ks, p.mw);
  return HRP_ERR_ARG;
}

}  // namespace hrp
