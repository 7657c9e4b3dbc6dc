// Fused backward of the row-strip convolutions (include/hrp.h, hrp_rowbw_*): data gradient + weight gradient of a 3x3 C -> C
// layer (C = 32 @ W = 64, C = 64 @ W = 32: the BasicBlock layers of the two high-resolution HRNet branches, reference
// HRnet.py:28-57, 80 % of the BasicBlock bytes of a training step) from ONE staging of the output gradient.
//
// What it replaces: hrp_conv2d_fwd (data gradient, conv_row.h, with its BatchNorm prologue writing the BatchNorm-input gradient
// as a side output) + hrp_conv2d_bwd_weight (which re-read that side output and the layer's forward input).  Per 8-row strip
// the separate kernels moved 240 KB (conv2 of a block) / 272 KB (conv1); this one moves 155 / 217 KB + its share of a slab.
//
// Structure (differs from conv_row.h where the roofline said so: those kernels reach 3.3 TB/s because a workgroup's phases -
// stage, transform, multiply, store - run one after the other with two workgroups per CU to overlap them):
//   * ONE persistent workgroup per CU (4 waves, one per SIMD, up to 512 registers each), walking a contiguous range of strips;
//     the strip tiles are DOUBLE BUFFERED in LDS (2 x 74 KB): the DMA of strip s + 1 is issued right after the barrier that
//     opens strip s and lands under its ~290 MFMAs per wave; the second prologue operand (BatchNorm input, through
//     registers) of strip s + 1 is requested before the epilogue of strip s.
//   * tile G  (10 rows incl. halo, the staged operand of the data gradient = dY after the BatchNorm-backward prologue) and
//     tile X8 (the 8 centre rows of the layer's forward input, optionally relu(bn(.)) in place) sit side by side.
//   * data gradient: the MFMA loop and the register epilogue of conv_row.h (weights as A fragments, re-read from L2 per strip so
//     that they are not live across the weight-gradient loop).
//   * weight gradient: dW[ky][kx] (32 x 32 blocks) += G^T[rho][x - kx + 1] * X8[rho - 2 + ky][x]; both operands are gathered with
//     ds_read_b64_tr_b16 (K = pixels, NHWC keeps channels contiguous); one G fragment feeds the three kernel rows that see it.
//     C = 64: wave = one (cout block, cin block) pair, all pixels; C = 32: wave = one 16-pixel column block of every row, the
//     four partial sums are combined through LDS once per workgroup.  Accumulators (9 x 16 registers) live across strips.
//   * launch = up to 4 problems; the launch's strips are split EVENLY over the workgroups (a workgroup may finish one problem
//     and start the next): no tail from 512 + 256 strips not dividing by 3.
// LDS swizzle: 16-byte slot' = slot ^ gsw(x) of the image column, applied on the DMA source address and on every read.  C = 64
// uses a different permutation than conv_row.h: the transpose reads of a 32-lane group touch 32 channels (half a pixel) of 4
// consecutive pixels, and pixels x, x + 2 share a 128-byte half of the bank row - bit 2 of gsw alternates with x >> 1 so that
// the two land in different 64-byte quarters (ds_read_b128 of the data-gradient loop only needs gsw to be a bijection of x >> 1).
#include "conv_row.h"
#include <string.h>

namespace hrp {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifndef HRP_ROWBW_WF_LEAD
#define HRP_ROWBW_WF_LEAD 12      // C = 64: weight-gradient steps (of 60) that run under the data gradient's weight loads
#endif

template <int C>
struct BwCfg {
  using R = RowCfg<C>;
  static constexpr int W = R::W, P = R::P, S = R::S, KS = R::KS, TH = 8, ROWB = R::ROWB, NROWS = 10, PXP = R::PXP;
  static constexpr int G_BYTES = (P + NROWS * ROWB + 255) & ~255;
  static constexpr int X_ROWB = W * P;                       // 4 096: dense rows, no padding pixel (never read shifted)
  static constexpr int BUF_BYTES = G_BYTES + TH * X_ROWB;
  static constexpr int CTAB_OFF = 2 * BUF_BYTES;             // [10][C] floats, rows as in conv_row.h
  static constexpr int STAT_OFF = CTAB_OFF + 10 * C * 4;     // [4 waves][64] floats
  static constexpr int LDS_BYTES = STAT_OFF + 4 * 64 * 4;
  static constexpr int NJ = W / 16;                          // 16-pixel k-steps per image row
  static constexpr int NJW = C == 32 ? 1 : 2;                // ... per wave
  static_assert(4 * 9 * 4096 <= 2 * BUF_BYTES, "the C = 32 cross-wave combine fits the tile buffers");
  __device__ static __forceinline__ int gsw(int x) {
    if (C == 32) return (x >> 2) & 3;
    const int k = (x >> 1) & 7;
    return ((k & 1) << 2) | (k >> 1);
  }
};

__device__ __forceinline__ bf16x8 tr_frag(const char* lo, const char* hi) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)lo);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)hi);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One segment: strips [s_lo, s_hi) of problem q (strip numbering of the problem: image-major, 8 rows each); slab = index of
// this workgroup's partial slab of the problem.
template <int C>
__device__ __forceinline__ void rowbw_body(const hrp_rowbw_desc& q, const RowPlan& rp, const int s_lo, const int s_hi,
                                           const int slab, const int stat_slot) {
  using B = BwCfg<C>;
  constexpr int W = B::W, P = B::P, S = B::S, KS = B::KS, TH = B::TH, ROWB = B::ROWB, NROWS = B::NROWS;
  constexpr int MT = C / 32, NCOL = W / 32;
  const hrp_conv_desc& d = q.conv;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ctab = (float*)(smem + B::CTAB_OFF);
  float* stat_lds = (float*)(smem + B::STAT_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int sel = wave & 1, rg = wave >> 1;
  const int col = NCOL == 2 ? sel : 0;     // data gradient: which 32-pixel column block of the rows
  const int m = MT == 2 ? sel : 0;         // ... which 32-channel output tile
  const int H = d.H;
  const int pro = d.pro_mode;
  const bool bnb = d.bnb_x != nullptr;
  const bool xact = q.wg_act != 0;

  // ---- staging: piece `wave` of every row; the lane's 16 bytes = (pixel lane / S of the piece, slot lane % S), holding the
  // LOGICAL slot (lane % S) ^ gsw(x)
  const int px_in_piece = lane / S;
  const int xcol = wave * B::PXP + px_in_piece;
  const int lslot = (lane % S) ^ B::gsw(xcol);
  const unsigned lane_off = (unsigned)(wave * 1024 + px_in_piece * P + lslot * 16);
  auto strip_of = [&](int s, int& n, int& y0) { n = fdiv(s, rp.fd_spi); y0 = (s - n * rp.spi) * TH; };
  auto stage = [&](int s, char* buf) {
    int n, y0;
    strip_of(s, n, y0);
    const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
    const char* xg = (const char*)d.x + img_off + lane_off;
    char* rows = buf + P;
#pragma unroll
    for (int rs = 0; rs < NROWS; ++rs) {
      const int y = y0 - 1 + rs;
      char* dst = rows + rs * ROWB + wave * 1024;
      if (y >= 0 && y < H) dma16(xg + y * (W * P), dst);
      else *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
    }
    const char* wg = (const char*)q.wg_x + img_off + lane_off + y0 * (W * P);
    char* xr = buf + B::G_BYTES + wave * 1024;
#pragma unroll
    for (int r = 0; r < TH; ++r) dma16(wg + r * (W * P), xr + r * B::X_ROWB);
  };
  // second prologue operand (pro_mode 2) of a strip through registers: issued a phase ahead of its use
  uint4 x2n[NROWS];
  int bitsn[NROWS];
  auto load_x2 = [&](int s) {
    int n, y0;
    strip_of(s, n, y0);
    const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
    const char* x2g = (const char*)d.pro_x2 + img_off + lane_off;
    const uint8_t* mg = d.pro_mask ? d.pro_mask + ((img_off + lane_off) >> 4) : nullptr;
#pragma unroll
    for (int rs = 0; rs < NROWS; ++rs) {
      const int y = y0 - 1 + rs;
      x2n[rs] = make_uint4(0, 0, 0, 0);
      bitsn[rs] = -1;
      if (y >= 0 && y < H) {
        x2n[rs] = *(const uint4*)(x2g + y * (W * P));
        if (mg) bitsn[rs] = mg[y * (W * P / 16)];
      }
    }
  };

  // (an empty range is legal and does nothing: the kernel calls both problems' bodies unconditionally - with the body inside a
  // branch the register allocator spilled 260 more registers)
  const bool any = s_lo < s_hi;
  if (any) stage(s_lo, smem);
  if (any && pro == 2) load_x2(s_lo);
  // the zero pixels of both buffers: one in front of row slot 0, one behind every row slot
  if (tid < 2 * (NROWS + 1) * S) {
    const int bsel = tid / ((NROWS + 1) * S), e = tid - bsel * (NROWS + 1) * S;
    const int k = e / S, j = e - k * S;
    *(uint4*)(smem + bsel * B::BUF_BYTES + (k == 0 ? 0 : P + (k - 1) * ROWB + W * P) + j * 16) = make_uint4(0, 0, 0, 0);
  }
  // per-channel constants (LDS table [10][C], conv_row.h): 0 sc, 1 sh of pro_stats; 2 a, 3 b, 4 k0, 5 k1 (prologue 2);
  // 6 a, 7 b, 8 sc, 9 sh of bnb_stats (epilogue reduce AND the activation of the weight gradient's X operand)
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
  if ((bnb || xact) && tid >= 64 && tid < 64 + C) {
    const int c = tid - 64;
    float mean, inv, sc, sh;
    row_bn_consts(d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps, c, C, mean, inv, sc, sh);
    ctab[8 * C + c] = sc; ctab[9 * C + c] = sh;
    ctab[6 * C + c] = inv; ctab[7 * C + c] = -mean * inv;
  }
  __syncthreads();                                                        // the constant table

  // ---- data gradient: weight rows of the lane (MFMA row -> channel permutation of conv_row.h) and B read addresses
  const int co_lane = m * 32 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  const char* wl = (const char*)d.w + co_lane * ROW + half * 16;
  int baddr[3][KS];
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi) {
    const int xq = col * 32 + l31 + dxi - 1;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      baddr[dxi][kk] = P + rg * 4 * ROWB + xq * P + (((2 * kk + half) ^ B::gsw(xq)) << 4);
  }
  const int cl = m * 32 + 16 * half;      // first output channel of the lane

  // ---- weight gradient: lane roles of the transpose reads.  Source lane (16-lane group grp, pixel tp of 4, channel quad qd)
  // supplies channels 16 grp + 4 qd .. + 3 of pixel tp; the destination lane l31 receives channel l31 of 4 pixels.  A lane's
  // K slots: pixels 16 j + 8 half + {0 .. 3} (first read) and + {4 .. 7} (second read).
  const int cob = C == 64 ? (wave >> 1) : 0, cib = C == 64 ? (wave & 1) : 0;
  const int jbase = C == 32 ? wave : 0;
  int offA[3][2], offB[2];
  {
    const int tp = (lane & 15) >> 2, qd = lane & 3, grp = (lane >> 4) & 1;
    const int sa = cob * 4 + 2 * grp + (qd >> 1), sb = cib * 4 + 2 * grp + (qd >> 1);
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
      const int xr = 8 * half + tp + 4 * hi;
      offB[hi] = B::G_BYTES + (16 * jbase + xr) * P + ((sb ^ B::gsw(xr)) << 4) + 8 * (qd & 1);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xa = xr - (kx - 1);             // -1 and 16 alias the neighbouring block's pixels / the shared zero pixels
        offA[kx][hi] = P + (16 * jbase + xa) * P + ((sa ^ B::gsw(xa)) << 4) + 8 * (qd & 1);
      }
    }
  }
  f32x16 wacc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) wacc[t][i] = 0.f;

  float vtot = 0.f;                       // statistics of the strips so far, already reduced over the half wave
  int it = 0;
  for (int s = s_lo; s < s_hi; ++s, ++it) {
    char* buf = smem + (it & 1) * B::BUF_BYTES;
    char* lds_rows = buf + P;
    int n, y0;
    strip_of(s, n, y0);
    const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this lane's DMA pieces (and x2n) of strip s have landed
    // ---- prologue: transform the bytes this lane staged, in place
    if (pro != 0) {
      RowPro pc;
      int cb = lslot * 8;                                       // the lane's 8 channels
      asm volatile("" : "+v"(cb));
      pc.load(ctab, C, cb);
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
        pc.load2(ctab, C, cb);
#pragma unroll
        for (int rs = 0; rs < NROWS; ++rs) {
          const int y = y0 - 1 + rs;
          if (y < 0 || y >= H) continue;
          char* p = lds_rows + rs * ROWB + wave * 1024 + lane * 16;
          uint4 gm;
          const uint4 o = pc.template bwd<true>(*(const uint4*)p, x2n[rs], bitsn[rs], gm);
          *(uint4*)p = o;
          if (rs >= 1 && rs <= TH) {
            const unsigned off = img_off + lane_off + y * (W * P);
            if (side) *(uint4*)(side + off) = o;
            if (d.pro_side2) row_side2(d, off, gm);
          }
        }
      }
    }
    if (xact) {                                               // X8 = relu(bn(wg_x)), the forward prologue's arithmetic
      RowPro pa;
      int cb = lslot * 8;
      asm volatile("" : "+v"(cb));
      pa.load(ctab + 8 * C, C, cb);                           // rows 8 / 9: sc, sh of bnb_stats
      char* xr = buf + B::G_BYTES + wave * 1024 + lane * 16;
#pragma unroll
      for (int r = 0; r < TH; ++r) *(uint4*)(xr + r * B::X_ROWB) = pa.act(*(const uint4*)(xr + r * B::X_ROWB));
    }
    __syncthreads();      // tiles of strip s complete; every wave is done with the other buffer (strip s - 1)

    // weights of the data gradient: A fragments of this wave's 32 output channels, from L2, requested inside the weight-gradient
    // loop (at step WF_AT: early enough to land under it, late enough - C = 64: 144 registers - not to be live next to its operands)
    bf16x8 wf[9][KS];
    auto load_wf = [&]() {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          wf[t][kk] = *(const bf16x8*)(wl + (size_t)((kk * d.w_ntaps + rp.wslot[t]) * C) * ROW);
    };
    if (s + 1 < s_hi) stage(s + 1, smem + ((it + 1) & 1) * B::BUF_BYTES);

    // ---- weight gradient.  Step (rho, jj, kx): G row slot rho (image row y0 - 1 + rho), k-step jj, tap column kx; the G fragment
    // feeds kernel rows ky = 0 .. 2 with the X8 fragments of rows rho - 2 + ky.  Fragments are read AHEAD steps before their MFMAs.
    {
      constexpr int NJW = B::NJW, NST = NROWS * NJW * 3, AHEAD = 2, RING = 4;
      constexpr int WF_AT = C == 32 ? 0 : NST - HRP_ROWBW_WF_LEAD;
      bf16x8 af[RING], bw[4][NJW];
      auto issue = [&](int t) {      // t is a constant after unrolling
        const int kx = t % 3, jj = (t / 3) % NJW, rho = t / (3 * NJW);
        if (kx == 0 && rho < TH) bw[rho & 3][jj] = tr_frag(buf + rho * B::X_ROWB + jj * 16 * P + offB[0], buf + rho * B::X_ROWB + jj * 16 * P + offB[1]);
        af[t % RING] = tr_frag(buf + rho * ROWB + jj * 16 * P + offA[kx][0], buf + rho * ROWB + jj * 16 * P + offA[kx][1]);
      };
#pragma unroll
      for (int t = 0; t < AHEAD; ++t) issue(t);
#pragma unroll
      for (int t = 0; t < NST; ++t) {
        if (t == WF_AT) load_wf();
        if (t + AHEAD < NST) issue(t + AHEAD);
        const int kx = t % 3, jj = (t / 3) % NJW, rho = t / (3 * NJW);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int r = rho - 2 + ky;
          if (r >= 0 && r < TH)
            wacc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t % RING], bw[r & 3][jj], wacc[ky * 3 + kx], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- data gradient MFMA loop (conv_row.h): input rows rg*4 - 1 .. rg*4 + 4 of the strip
    f32x16 acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[o][i] = 0.f;
    {
      constexpr int NSTEP = 6 * 3 * KS, RING = 4, AHEAD = 3;
      bf16x8 bq[RING];
      auto rd = [&](int qq) -> bf16x8 {
        const int irel = qq / (3 * KS), dxi = (qq / KS) % 3, kk = qq % KS;
        return *(const bf16x8*)(buf + baddr[dxi][kk] + irel * ROWB);
      };
#pragma unroll
      for (int qq = 0; qq < AHEAD; ++qq) bq[qq % RING] = rd(qq);
#pragma unroll
      for (int qq = 0; qq < NSTEP; ++qq) {
        if (qq + AHEAD < NSTEP) bq[(qq + AHEAD) % RING] = rd(qq + AHEAD);
        const int irel = qq / (3 * KS), dxi = (qq / KS) % 3, kk = qq % KS;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int dyi = irel - o;
          if (dyi >= 0 && dyi <= 2)
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[dyi * 3 + dxi][kk], bq[qq % RING], acc[o], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the next strip's second operand: requested now, consumed after the epilogue
    if (pro == 2 && s + 1 < s_hi) load_x2(s + 1);

    // ---- epilogue: lane = pixel (row y0 + rg*4 + o, x = col*32 + l31), channels m*32 + 16*half .. +15
    float s1[16], s2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
    {
      const unsigned out_off = img_off + (unsigned)((y0 + rg * 4) * W + col * 32 + l31) * P + cl * 2;
      constexpr int EG = 2;
#pragma unroll
      for (int o0 = 0; o0 < 4; o0 += EG) {
        unsigned off[EG];
#pragma unroll
        for (int o = 0; o < EG; ++o) off[o] = out_off + (o0 + o) * (W * P);
        row_epilogue<EG, true>(d, *(const f32x16(*)[EG])&acc[o0], off, (1u << EG) - 1, cl, ctab, C, bnb, s1, s2);
      }
    }
    if (d.stats) vtot += row_reduce32(s1, s2, l31);
  }

  // ---- statistics of the segment (conv_row.h)
  if (d.stats && any) {
    stat_lds[wave * 64 + lane] = vtot;
    __syncthreads();
    if (tid < 2 * C) {
      const int which = tid / C, c = tid - which * C;
      const int mc = c >> 5, hq = (c >> 4) & 1, j = c & 15;
      auto tot = [&](int wh) {
        const int li = hq * 32 + wh * 16 + j;
        float t = 0.f;
        if (MT == 1) t = (stat_lds[0 * 64 + li] + stat_lds[1 * 64 + li]) + (stat_lds[2 * 64 + li] + stat_lds[3 * 64 + li]);
        else t = stat_lds[mc * 64 + li] + stat_lds[(mc + 2) * 64 + li];
        return t;
      };
      float t = tot(which);
      if (bnb && which == 1) t = fmaf(ctab[6 * C + c], t, ctab[7 * C + c] * tot(0));     // sum g * xhat = a * sum g x + b * sum g
      atomicAdd(d.stats + stat_slot * 2 * C + which * C + c, (double)t);
    }
  }

  // ---- partial slab of the weight gradient: ws[slab][block pair][tap][cout row 32][cin 32] (hrp_wgrad_fold_desc)
  // accumulator register i of lane (l31, half): row (i & 3) + 8 (i >> 2) + 4 half, column l31
  constexpr int PAIRS = MT * MT;
  float* ws = (float*)q.workspace + (size_t)slab * (PAIRS * 9 * 1024);
  if (!any) return;
  if (C == 64) {
    float* mine = ws + (size_t)(cob * 2 + cib) * (9 * 1024) + 4 * half * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[t * 1024 + ((i & 3) + 8 * (i >> 2)) * 32] = wacc[t][i];
  } else {
    __syncthreads();                        // every wave is done with the tiles
    float* dump = (float*)smem;
    float* mine = dump + wave * (9 * 1024) + 4 * half * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[t * 1024 + ((i & 3) + 8 * (i >> 2)) * 32] = wacc[t][i];
    __syncthreads();
    for (int f = tid; f < 9 * 256; f += 256) {       // fixed order: wave 0 + 1 + 2 + 3
      float4 v = ((const float4*)dump)[f];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 u = ((const float4*)(dump + w * (9 * 1024)))[f];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      ((float4*)ws)[f] = v;
    }
  }
}

struct RowBwArgs {
  hrp_rowbw_desc q[HRP_ROWBW_MAX];
  RowPlan rp[HRP_ROWBW_MAX];
  int strip0[HRP_ROWBW_MAX + 1];
  int first_wg[HRP_ROWBW_MAX];
  int n, total, nwg, pad;
};

// Problems are addressed with COMPILE-TIME indices into the by-value kernel argument: descriptor fields then are scalar loads
// from the kernarg segment that the compiler re-issues instead of keeping them live (with a run-time problem index, or a table in
// global memory, it kept - and spilled - whole descriptors: 5 000 spilled SGPRs, 1.4 KiB of scratch per lane).
// C0 / C1: channel counts of problem 0 / 1 (C1 == 0: one problem).
template <int C0, int C1>
__global__ __launch_bounds__(256, 1) void rowbw_kernel(const RowBwArgs A) {
  const int w = blockIdx.x;
  const int lo = (int)((long long)w * A.total / A.nwg), hi = (int)((long long)(w + 1) * A.total / A.nwg);
  const int b0 = hi < A.strip0[1] ? hi : A.strip0[1];
  rowbw_body<C0>(A.q[0], A.rp[0], lo, b0, w - A.first_wg[0], w & (HRP_STAT_SLOTS - 1));
  if constexpr (C1 != 0) {
    const int a1 = lo > A.strip0[1] ? lo : A.strip0[1];
    __syncthreads();      // (a workgroup that crosses the boundary: the second segment re-initialises the tiles)
    rowbw_body<C1>(A.q[1], A.rp[1], a1 - A.strip0[1], hi - A.strip0[1], w - A.first_wg[1], w & (HRP_STAT_SLOTS - 1));
  }
}

static int rowbw_channels(const hrp_rowbw_desc& q) {
  const int C = row_channels(q.conv);
  if (C != 32 && C != 64) return 0;
  if (!q.wg_x || !q.dw || (uintptr_t)q.wg_x % 16 || (uintptr_t)q.dw % 16) return 0;
  if (q.wg_act && (!q.conv.bnb_stats || !q.conv.bnb_gamma || !q.conv.bnb_beta)) return 0;
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
  int total = 0, lds = 0;
  HRP_REQUIRE(n == 1 || (rowbw_channels(descs[0]) == 32 && rowbw_channels(descs[1]) == 64),
              "rowbw: two problems of one launch are a 32-channel and a 64-channel one, in this order");
  for (int i = 0; i < n; ++i) {
    const int C = rowbw_channels(descs[i]);
    HRP_REQUIRE(C != 0, "rowbw: problem %d is not a 32- / 64-channel row-strip data gradient with wg_x / dw set", i);
    const int rc = conv_check(&descs[i].conv);
    if (rc != HRP_OK) return rc;
    info->strip0[i] = total;
    total += descs[i].conv.N * (descs[i].conv.H / 8);
    const int l = C == 32 ? BwCfg<32>::LDS_BYTES : BwCfg<64>::LDS_BYTES;
    lds = l > lds ? l : lds;
  }
  info->strip0[n] = total;
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
  const int nwg = total < max_wgs ? total : max_wgs;
  info->n = n; info->grid = nwg; info->lds_bytes = lds; info->total_strips = total;
  for (int i = 0; i < n; ++i) {
    // workgroup w holds strips [w * total / nwg, (w + 1) * total / nwg)
    int first = -1, last = -1;
    for (int w = 0; w < nwg; ++w) {
      const int lo = (int)((long long)w * total / nwg), hi = (int)((long long)(w + 1) * total / nwg);
      if (lo < info->strip0[i + 1] && hi > info->strip0[i]) { if (first < 0) first = w; last = w; }
    }
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
      A.strip0[i] = info->strip0[i];
    }
    A.strip0[n] = total;
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
  HRP_REQUIRE(A.n == info->n && A.nwg == info->grid && A.total == info->total_strips, "rowbw launch: table and info do not belong together");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)rowbw_kernel<32, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)rowbw_kernel<64, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)rowbw_kernel<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int c0 = A.q[0].conv.Cin, c1 = A.n == 2 ? A.q[1].conv.Cin : 0;
  const dim3 grid(info->grid), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (c0 == 32 && c1 == 64) hipLaunchKernelGGL((rowbw_kernel<32, 64>), grid, block, info->lds_bytes, s, A);
  else if (c0 == 32 && c1 == 0) hipLaunchKernelGGL((rowbw_kernel<32, 0>), grid, block, info->lds_bytes, s, A);
  else if (c0 == 64 && c1 == 0) hipLaunchKernelGGL((rowbw_kernel<64, 0>), grid, block, info->lds_bytes, s, A);
  else { set_error("rowbw launch: channel combination (%d, %d)", c0, c1); return HRP_ERR_ARG; }
  return check_launch("rowbw_kernel");
}
