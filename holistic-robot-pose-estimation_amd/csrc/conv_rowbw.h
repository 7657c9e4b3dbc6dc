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
#pragma once
#include "conv_row.h"
#include <string.h>

namespace hrp {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifndef HRP_ROWBW_STAGGER
#define HRP_ROWBW_STAGGER 1       // role 1 runs its MFMA loop after B2 (beside role 0's epilogue) instead of before it
#endif
#ifndef HRP_ROWBW_EO_EARLY
#define HRP_ROWBW_EO_EARLY 1      // role 0 requests its epilogue operands before B2 instead of behind it
#endif
#ifndef HRP_ROWBW_WF_LEAD
#define HRP_ROWBW_WF_LEAD 12      // C = 64: weight-gradient steps (of 60) that run under the data gradient's weight loads
#endif

// -DHRP_TIMELINE (development build only): thread 0 of every workgroup stamps the 100 MHz wall clock at the phase boundaries of
// its first two strips; tools/bench_rowbw.py --timeline reads them with hrp_debug_rowbw_timeline.
#ifdef HRP_TIMELINE
static __device__ unsigned long long g_rowbw_timeline[1024 * 32];
#define HRP_BSTAMP(i) do { if (tid == 256 && blockIdx.x < 1024 && it < 2) g_rowbw_timeline[blockIdx.x * 32 + it * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define HRP_ASTAMP(i) do { if (tid == 0 && blockIdx.x < 1024 && it < 2) g_rowbw_timeline[blockIdx.x * 32 + 16 + it * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HRP_BSTAMP(i) do { } while (0)
#define HRP_ASTAMP(i) do { } while (0)
#endif

template <int C>
struct BwCfg {
  using R = RowCfg<C>;
  static constexpr int W = R::W, P = R::P, S = R::S, KS = R::KS, TH = 8, ROWB = R::ROWB, NROWS = 10, PXP = R::PXP;
  static constexpr int G_BYTES = (P + NROWS * ROWB + 255) & ~255;
  static constexpr int X_ROWB = W * P;                       // 4 096: dense rows, no padding pixel (never read shifted)
  static constexpr int BUF_BYTES = G_BYTES + TH * X_ROWB;
  static constexpr int CTAB_OFF = 2 * BUF_BYTES;             // [10][C] floats, rows as in conv_row.h
  static constexpr int STAT_OFF = CTAB_OFF + 10 * C * 4;     // [4 waves][64 lanes][2] floats
  static constexpr int LDS_BYTES = STAT_OFF + 4 * 64 * 2 * 4;
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

// Reduce-scatter of the lane's 32 partial sums (s1[0 .. 15], s2[0 .. 15]) over its ROW of 16 lanes with DPP lane exchanges only
// (row_mirror, row_half_mirror, two quad permutations: each pairs a lane with one whose kept half is the other one).  No LDS
// instruction: conv_row.h's row_reduce32 uses ds_bpermute, whose results came back wrong in two lanes, run-to-run different,
// while the other role's waves streamed ds_read_b64_tr_b16 on the same CU (tools/dbg_rowbw_race.py).
// -> lane L of the row (0 .. 15): L < 8: totals of s1[2L], s1[2L + 1]; L >= 8: totals of s2[2(L - 8)], s2[2(L - 8) + 1].
template <int CTRL>
__device__ __forceinline__ float bw_dpp(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float2 bw_reduce16_dpp(const float (&s1)[16], const float (&s2)[16], const int lane) {
  float v16[16], v8[8], v4[4], v2[2];
  {
    const bool up = (lane & 8) != 0;                 // row_mirror: lane i <-> 15 - i
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float keep = up ? s2[j] : s1[j], send = up ? s1[j] : s2[j]; v16[j] = keep + bw_dpp<0x140>(send); }
  }
  {
    const bool up = (lane & 4) != 0;                 // row_half_mirror: i <-> 7 - i within 8
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float keep = up ? v16[8 + j] : v16[j], send = up ? v16[j] : v16[8 + j]; v8[j] = keep + bw_dpp<0x141>(send); }
  }
  {
    const bool up = (lane & 2) != 0;                 // quad_perm [3, 2, 1, 0]
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float keep = up ? v8[4 + j] : v8[j], send = up ? v8[j] : v8[4 + j]; v4[j] = keep + bw_dpp<0x1b>(send); }
  }
  {
    const bool up = (lane & 1) != 0;                 // quad_perm [1, 0, 3, 2]
#pragma unroll
    for (int j = 0; j < 2; ++j) { const float keep = up ? v4[2 + j] : v4[j], send = up ? v4[j] : v4[2 + j]; v2[j] = keep + bw_dpp<0xb1>(send); }
  }
  return make_float2(v2[0], v2[1]);
}

// One segment: strips [s_lo, s_hi) of problem q (strip numbering of the problem: image-major, 8 rows each); slab = index of
// this workgroup's partial slab of the problem.  512 threads: waves 0-3 = data-gradient role (all VALU-heavy work: no state
// lives across strips there), waves 4-7 = weight-gradient role (its 144 accumulator registers live across the strips).
//
//   data-gradient waves (role 0)                    weight-gradient waves (role 1)
//   --------------------------------------------    ---------------------------------------------------
//   request epilogue operands, rows 0-4 of the      weight-gradient MFMA loop on buffer b (strip s)
//   next prologue's second operand, weights (L2)    wait: DMA of strip s + 1 (issued one strip ago) landed
//   MFMA loop on buffer b
//   -------------------------------------------- B2: both MFMA loops are done with buffer b; buffer b ^ 1 has landed
//   epilogue of strip s (BatchNorm sums, stores)    DMA of strip s + 2 -> buffer b
//   prologue of strip s + 1 in place on the G       X8 tile of strip s + 1: relu(bn(.)) in place (wg_act)
//   tile of buffer b ^ 1 (BatchNorm backward)
//   -------------------------------------------- B1: buffer b ^ 1 holds strip s + 1
// Each SIMD hosts one wave of either role: matrix work of one overlaps VALU / memory work of the other.  The DMA engine is
// driven by role 1 only: its waves issue no compiler-visible global loads, so no compiler-placed s_waitcnt vmcnt ever drains the
// pieces in flight (the wait-count pass cannot see the inline-asm DMA and would wait for everything older than its own loads).
// FORM: which options of the descriptor are compile-time facts of the kernel (host: rowbw_form()).  One form per kernel: with
// several epilogue / prologue versions behind workgroup-uniform branches in ONE kernel the register allocator spilled 160 - 600
// values, some inside the MFMA loops (a single version: none).
//   0  generic: every option decided at run time (uniform branches)
//   1  conv2 of a block: prologue mask as bits (pro_mask), epilogue reduce with the mask recomputed, no residual, no side outputs
//   2  conv1 of a block: prologue mask recomputed, residual (optionally under res_mask), no statistics, no side outputs
//   3  conv1 of a block + the previous block's reduce: as 2 with an epilogue reduce whose mask comes as bits (bnb_mask)
template <int C, int FORM>
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
  const int role = wave >> 2, rw = wave & 3;      // role 0: data gradient, 1: weight gradient; rw: wave of the role
  const int l31 = lane & 31, half = lane >> 5;
  const int H = d.H;
  const int pro = d.pro_mode;
  const bool bnb = d.bnb_x != nullptr;
  const bool xact = q.wg_act != 0;
  const bool ub = FORM == 0 ? d.pro_mask != nullptr : FORM == 1, wgm = FORM == 0 && d.pro_side2 != nullptr;     // (uniform: RowPro::bwd)
  // (an empty range is legal and does nothing: the kernel calls both problems' bodies unconditionally - with the body inside a
  // branch the register allocator spilled 260 more registers)
  const bool any = s_lo < s_hi;
  auto strip_of = [&](int s, int& n, int& y0) { n = fdiv(s, rp.fd_spi); y0 = (s - n * rp.spi) * TH; };
  // a role's 256 lanes cover one 4 KiB row: lane = piece `rw` of the row, 16 bytes = (pixel lane / S of the piece, slot lane % S),
  // holding the LOGICAL slot (lane % S) ^ gsw(x).  Role 1 stages with this map, role 0 transforms the G tile with it.
  const int px_in_piece = lane / S;
  const int xcol = rw * B::PXP + px_in_piece;
  const int lslot = (lane % S) ^ B::gsw(xcol);
  const unsigned lane_off = (unsigned)(rw * 1024 + px_in_piece * P + lslot * 16);

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
  if ((bnb || xact) && tid >= 256 && tid < 256 + C) {
    const int c = tid - 256;
    float mean, inv, sc, sh;
    row_bn_consts(d.bnb_stats, d.bnb_gamma, d.bnb_beta, d.bnb_count, d.bnb_eps, c, C, mean, inv, sc, sh);
    ctab[8 * C + c] = sc; ctab[9 * C + c] = sh;
    ctab[6 * C + c] = inv; ctab[7 * C + c] = -mean * inv;
  }
  __syncthreads();                                                        // the constant table, the zero pixels

  if (role == 1) {
    // =================================================================================================================
    // weight-gradient role: the DMA of every strip, the activation of the X8 tile, the weight-gradient MFMA loops, the slab
    // =================================================================================================================
    auto stage = [&](int s, char* buf) {
      int n, y0;
      strip_of(s, n, y0);
      const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
      const char* xg = (const char*)d.x + img_off + lane_off;
      char* rows = buf + P;
#pragma unroll
      for (int rs = 0; rs < NROWS; ++rs) {
        const int y = y0 - 1 + rs;
        char* dst = rows + rs * ROWB + rw * 1024;
        if (y >= 0 && y < H) dma16(xg + y * (W * P), dst);
        else *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
      }
      const char* wg = (const char*)q.wg_x + img_off + lane_off + y0 * (W * P);
      char* xr = buf + B::G_BYTES + rw * 1024;
#pragma unroll
      for (int r = 0; r < TH; ++r) dma16(wg + r * (W * P), xr + r * B::X_ROWB);
    };
    // X8 = relu(bn(wg_x)), the forward prologue's arithmetic, in place
    auto act_x8 = [&](char* buf) {
      RowPro pa;
      int cb = lslot * 8;
      asm volatile("" : "+v"(cb));
      pa.load(ctab + 8 * C, C, cb);                           // rows 8 / 9: sc, sh of bnb_stats
      char* xr = buf + B::G_BYTES + rw * 1024 + lane * 16;
#pragma unroll
      for (int r = 0; r < TH; ++r) *(uint4*)(xr + r * B::X_ROWB) = pa.act(*(const uint4*)(xr + r * B::X_ROWB));
    };

    // lane roles of the transpose reads.  Source lane (16-lane group grp, pixel tp of 4, channel quad qd) supplies channels
    // 16 grp + 4 qd .. + 3 of pixel tp; the destination lane l31 receives channel l31 of 4 pixels.  A lane's K slots: pixels
    // 16 j + 8 half + {0 .. 3} (first read) and + {4 .. 7} (second read).
    const int cob = C == 64 ? (rw >> 1) : 0, cib = C == 64 ? (rw & 1) : 0;
    const int jbase = C == 32 ? rw : 0;
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

    if (any) {
      stage(s_lo, smem);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (xact) act_x8(smem);
    }
    __syncthreads();                                                      // Bp: the first strip has landed (role 0: its prologue)
    if (any && s_lo + 1 < s_hi) stage(s_lo + 1, smem + B::BUF_BYTES);
    __syncthreads();                                                      // B1 of the first strip
    int it = 0;
    for (int s = s_lo; s < s_hi; ++s, ++it) {
      char* buf = smem + (it & 1) * B::BUF_BYTES;
      char* obuf = smem + ((it + 1) & 1) * B::BUF_BYTES;
      HRP_BSTAMP(0);
#if !HRP_ROWBW_STAGGER
      // The MFMA loop runs AFTER B2, beside role 0's epilogue / prologue (VALU, memory) instead of beside its MFMA loop: two
      // matrix loops on one SIMD only take turns (C = 64: 2 x 144 MFMAs = 3.8 us of pipe time per strip).
      // Step (rho, jj, kx): G row slot rho (image row y0 - 1 + rho), k-step jj, tap column kx; the G fragment feeds kernel rows
      // ky = 0 .. 2 with the X8 fragments of rows rho - 2 + ky.  Fragments are read AHEAD steps before their MFMAs.
      {
        constexpr int NJW = B::NJW, NST = NROWS * NJW * 3, AHEAD = 2, RING = 4;
        // window of X8 row fragments: row rho is read AHEAD steps before step (rho, 0, 0), when row rho - 3 of the same k-step is
        // dead only if a row has more than AHEAD / 3 k-steps: 3 slots for C = 64, 4 for C = 32
        constexpr int BWS = NJW == 1 ? 4 : 3;
        bf16x8 af[RING], bw[BWS][NJW];
        auto issue = [&](int t) {      // t is a constant after unrolling
          const int kx = t % 3, jj = (t / 3) % NJW, rho = t / (3 * NJW);
          if (kx == 0 && rho < TH) bw[rho % BWS][jj] = tr_frag(buf + rho * B::X_ROWB + jj * 16 * P + offB[0], buf + rho * B::X_ROWB + jj * 16 * P + offB[1]);
          af[t % RING] = tr_frag(buf + rho * ROWB + jj * 16 * P + offA[kx][0], buf + rho * ROWB + jj * 16 * P + offA[kx][1]);
        };
#pragma unroll
        for (int t = 0; t < AHEAD; ++t) issue(t);
#pragma unroll
        for (int t = 0; t < NST; ++t) {
          if (t + AHEAD < NST) issue(t + AHEAD);
          const int kx = t % 3, jj = (t / 3) % NJW, rho = t / (3 * NJW);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int r = rho - 2 + ky;
            if (r >= 0 && r < TH)
              wacc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t % RING], bw[r % BWS][jj], wacc[ky * 3 + kx], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the DMA pieces of strip s + 1 (issued one strip ago)
      HRP_BSTAMP(1);
      __syncthreads();                                                    // B2: role 0 is done with `buf`; `obuf` has landed
      HRP_BSTAMP(2);
#if HRP_ROWBW_STAGGER
      // The MFMA loop runs AFTER B2, beside role 0's epilogue / prologue (VALU, memory) instead of beside its MFMA loop: two
      // matrix loops on one SIMD only take turns (C = 64: 2 x 144 MFMAs = 3.8 us of pipe time per strip).
      // Step (rho, jj, kx): G row slot rho (image row y0 - 1 + rho), k-step jj, tap column kx; the G fragment feeds kernel rows
      // ky = 0 .. 2 with the X8 fragments of rows rho - 2 + ky.  Fragments are read AHEAD steps before their MFMAs.
      {
        constexpr int NJW = B::NJW, NST = NROWS * NJW * 3, AHEAD = 2, RING = 4;
        // window of X8 row fragments: row rho is read AHEAD steps before step (rho, 0, 0), when row rho - 3 of the same k-step is
        // dead only if a row has more than AHEAD / 3 k-steps: 3 slots for C = 64, 4 for C = 32
        constexpr int BWS = NJW == 1 ? 4 : 3;
        bf16x8 af[RING], bw[BWS][NJW];
        auto issue = [&](int t) {      // t is a constant after unrolling
          const int kx = t % 3, jj = (t / 3) % NJW, rho = t / (3 * NJW);
          if (kx == 0 && rho < TH) bw[rho % BWS][jj] = tr_frag(buf + rho * B::X_ROWB + jj * 16 * P + offB[0], buf + rho * B::X_ROWB + jj * 16 * P + offB[1]);
          af[t % RING] = tr_frag(buf + rho * ROWB + jj * 16 * P + offA[kx][0], buf + rho * ROWB + jj * 16 * P + offA[kx][1]);
        };
#pragma unroll
        for (int t = 0; t < AHEAD; ++t) issue(t);
#pragma unroll
        for (int t = 0; t < NST; ++t) {
          if (t + AHEAD < NST) issue(t + AHEAD);
          const int kx = t % 3, jj = (t / 3) % NJW, rho = t / (3 * NJW);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int r = rho - 2 + ky;
            if (r >= 0 && r < TH)
              wacc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t % RING], bw[r % BWS][jj], wacc[ky * 3 + kx], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#endif
      HRP_BSTAMP(3);
      if (s + 2 < s_hi) stage(s + 2, buf);                                // (this role was the last reader of `buf`)
      HRP_BSTAMP(4);
      if (xact && s + 1 < s_hi) act_x8(obuf);
      HRP_BSTAMP(5);
      __syncthreads();                                                    // B1: strip s + 1 is ready in `obuf`
      HRP_BSTAMP(6);
    }

    // ---- partial slab of the weight gradient: ws[slab][block pair][tap][cout row 32][cin 32] (hrp_wgrad_fold_desc)
    // accumulator register i of lane (l31, half): row (i & 3) + 8 (i >> 2) + 4 half, column l31
    constexpr int PAIRS = MT * MT;
    float* ws = (float*)q.workspace + (size_t)slab * (PAIRS * 9 * 1024);
    if (C == 64) {
      if (any) {
        float* mine = ws + (size_t)(cob * 2 + cib) * (9 * 1024) + 4 * half * 32 + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) mine[t * 1024 + ((i & 3) + 8 * (i >> 2)) * 32] = wacc[t][i];
      }
    } else {
      float* mine = (float*)smem + rw * (9 * 1024) + 4 * half * 32 + l31;      // (every wave is past the last B1: the tiles are free)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) mine[t * 1024 + ((i & 3) + 8 * (i >> 2)) * 32] = wacc[t][i];
    }
  } else {
    // =================================================================================================================
    // data-gradient role: the prologue of the G tile, the MFMA loop and the register epilogue of conv_row.h
    // =================================================================================================================
    const int sel = rw & 1, rg = rw >> 1;
    const int col = NCOL == 2 ? sel : 0;     // which 32-pixel column block of the rows
    const int m = MT == 2 ? sel : 0;         // which 32-channel output tile
    // weight rows of the lane (MFMA row -> channel permutation of conv_row.h) and B read addresses
    const int co_lane = m * 32 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    // weights through buffer loads: ONE lane offset register for all fragments, the fragment's offset in an SGPR (with global
    // loads the compiler kept a 64-bit address per fragment - 72 registers for the 64-channel kernel - and spilled them)
    const unsigned lane_w = (unsigned)(co_lane * ROW + half * 16);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, 0x7fffffff, 0x00020000);
    // B read addresses: K chunk kk adds kk << 5 by XOR (the pixel base is P-aligned and carries no bits below the slot field)
    int baddr0[3];
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int xq = col * 32 + l31 + dxi - 1;
      baddr0[dxi] = P + rg * 4 * ROWB + xq * P + ((half ^ B::gsw(xq)) << 4);
    }
    const int cl = m * 32 + 16 * half;      // first output channel of the lane
    float vtot0 = 0.f, vtot1 = 0.f;         // statistics of the strips so far, reduced over the lane's row of 16 (bw_reduce16_dpp)

    // second prologue operand (pro_mode 2: the BatchNorm input, and the mask byte of the vector) of rows r0 .. r0 + 4 of a strip
    auto load_x2 = [&](int s, int r0, uint4 (&x2)[5], int (&bits)[5]) {
      int n, y0;
      strip_of(s, n, y0);
      const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
      const char* x2g = (const char*)d.pro_x2 + img_off;
      const uint8_t* mg = d.pro_mask + (img_off >> 4);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int y = y0 - 1 + r0 + j;
        x2[j] = make_uint4(0, 0, 0, 0);
        bits[j] = 0;
        if (y >= 0 && y < H) {
          x2[j] = *(const uint4*)(x2g + y * (W * P) + lane_off);
          // (the lane's mask byte out of an aligned dword load: byte loads moved 64 bytes per wave instruction and ten of them
          // took longer to come back than the 16-byte rows)
          if (ub) {
            const unsigned bi = (unsigned)(y * (W * P / 16)) + (lane_off >> 4);
            bits[j] = (int)((*(const unsigned*)(mg + (bi & ~3u)) >> (8 * (bi & 3))) & 0xffu);
          }
        }
      }
    };
    // prologue of the G tile: transform it in place (every lane the 16 bytes of its slot of every row; the tile has landed and
    // nobody reads it before the next B1)
    auto prologue = [&](int s, char* buf, const uint4 (&x2a)[5], const int (&bita)[5], const uint4 (&x2b)[5], const int (&bitb)[5]) {
      int n, y0;
      strip_of(s, n, y0);
      const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
      char* lds_rows = buf + P;
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
          char* p = lds_rows + rs * ROWB + rw * 1024 + lane * 16;
          const uint4 o = pc.act(*(const uint4*)p);
          *(uint4*)p = o;
          if (side && rs >= 1 && rs <= TH) *(uint4*)(side + img_off + lane_off + y * (W * P)) = o;
        }
      } else {
        pc.load2(ctab, C, cb);
        auto rows5 = [&](int r0, const uint4 (&x2)[5], const int (&bits)[5]) {
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int rs = r0 + j, y = y0 - 1 + rs;
            if (y < 0 || y >= H) continue;
            char* p = lds_rows + rs * ROWB + rw * 1024 + lane * 16;
            uint4 gm;
            uint4 o;
            if constexpr (FORM == 0) o = pc.template bwd<true>(*(const uint4*)p, x2[j], bits[j], gm, ub, wgm);
            else o = pc.template bwd_t<FORM == 1, false>(*(const uint4*)p, x2[j], bits[j], gm);
            *(uint4*)p = o;
            if constexpr (FORM == 0) {
              if (rs >= 1 && rs <= TH) {
                const unsigned off = img_off + lane_off + y * (W * P);
                if (side) *(uint4*)(side + off) = o;
                if (d.pro_side2) row_side2(d, off, gm);
              }
            }
          }
        };
        rows5(0, x2a, bita);
        rows5(5, x2b, bitb);
      }
    };
    // epilogue: the fast forms (1 - 3) request their tensor operands in front of the MFMA loop; form 0 runs conv_row.h's
    // generic epilogue (every option, loads inside)

    auto run = [&](auto fastc, auto bnbc, auto bitsc, auto resc) {
      constexpr bool FAST = decltype(fastc)::value, BNB = decltype(bnbc)::value, BITS = decltype(bitsc)::value, RES = decltype(resc)::value;
      uint4 x2a[5], x2b[5];      // the next prologue's second operand (rows 0 .. 4, 5 .. 9)
      int bita[5], bitb[5];
      bf16x8 wf[9];      // A fragments of this wave's 32 output channels for the current K chunk: tap dyi * 3 + dxi
      auto load_w3 = [&](int kk, int dyi) {
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi)
          wf[dyi * 3 + dxi] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane_w, ((kk * d.w_ntaps + rp.wslot[dyi * 3 + dxi]) * C) * ROW, 0));
      };
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) load_w3(0, dyi);
      __syncthreads();                                                      // Bp: the first strip has landed
      if (any && pro != 0) {
        if (pro == 2) { load_x2(s_lo, 0, x2a, bita); load_x2(s_lo, 5, x2b, bitb); }
        prologue(s_lo, smem, x2a, bita, x2b, bitb);
      }
      __syncthreads();                                                      // B1 of the first strip
      int it = 0;
      for (int s = s_lo; s < s_hi; ++s, ++it) {
        char* buf = smem + (it & 1) * B::BUF_BYTES;
        char* obuf = smem + ((it + 1) & 1) * B::BUF_BYTES;
        int n, y0;
        strip_of(s, n, y0);
        const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
        // lane = pixel (row y0 + rg*4 + o, x = col*32 + l31), channels m*32 + 16*half .. +15
        unsigned off[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) off[o] = img_off + (unsigned)((y0 + rg * 4 + o) * W + col * 32 + l31) * P + cl * 2;
        HRP_ASTAMP(0);
        // MFMA loop: input rows rg*4 - 1 .. rg*4 + 4 of the strip; step qq = (K chunk kk, input row irel, tap column dxi).
        // Weights: ONE buffer of 9 A fragments (36 registers), reloaded from L2 for chunk kk + 1 as the taps of chunk kk retire
        // (kernel row dyi is last used by input row irel = dyi + 3; its fragments are needed again at irel = dyi of the next
        // chunk: 6 steps = 12+ MFMAs later).  Chunk 0 of the next strip is requested behind the last step.
        f32x16 acc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[o][i] = 0.f;
        {
          constexpr int NSTEP = KS * 18, RING = 4, AHEAD = 3;
          bf16x8 bq[RING];
          auto rd = [&](int qq) -> bf16x8 {
            const int kk = qq / 18, irel = (qq % 18) / 3, dxi = qq % 3;
            return *(const bf16x8*)(buf + (baddr0[dxi] ^ (kk << 5)) + irel * ROWB);
          };
#pragma unroll
          for (int qq = 0; qq < AHEAD; ++qq) bq[qq % RING] = rd(qq);
#pragma unroll
          for (int qq = 0; qq < NSTEP; ++qq) {
            const int kk = qq / 18, irel = (qq % 18) / 3, dxi = qq % 3;
            if (qq + AHEAD < NSTEP) bq[(qq + AHEAD) % RING] = rd(qq + AHEAD);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
              const int dyi = irel - o;
              if (dyi >= 0 && dyi <= 2)
                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[dyi * 3 + dxi], bq[qq % RING], acc[o], 0, 0, 0);
            }
            // (the last chunk wraps to chunk 0: the next strip's - never issued without a next strip: a request still in flight at
            // the end of the loop came back INTO THE EPILOGUE'S REGISTERS - nondeterministic BatchNorm sums of one channel)
            if (dxi == 2 && irel >= 3 && (kk + 1 < KS || s + 1 < s_hi)) load_w3((kk + 1) % KS, irel - 3);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // Tensor operands of the epilogue, then of the next strip's prologue, requested only now: vmcnt counts in order, so a load
        // of the rolling weight buffer inside the MFMA loop would wait for every older request - these come from HBM - first.
        HRP_ASTAMP(1);
        RowEpiOps<4> eo;
#if HRP_ROWBW_EO_EARLY
        if constexpr (FAST) row_epi_load<4, BNB, BITS, RES>(d, off, 0xfu, eo);
        if (pro == 2 && s + 1 < s_hi) { load_x2(s + 1, 0, x2a, bita); load_x2(s + 1, 5, x2b, bitb); }
#endif
        HRP_ASTAMP(2);
        __syncthreads();                                                    // B2: both roles are done with `buf`; `obuf` has landed
        HRP_ASTAMP(3);
#if !HRP_ROWBW_EO_EARLY
        if constexpr (FAST) row_epi_load<4, BNB, BITS, RES>(d, off, 0xfu, eo);
        if (pro == 2 && s + 1 < s_hi) { load_x2(s + 1, 0, x2a, bita); load_x2(s + 1, 5, x2b, bitb); }
#endif
        float s1[16], s2[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
        if constexpr (!FAST) {
          constexpr int EG = 2;
#pragma unroll
          for (int o0 = 0; o0 < 4; o0 += EG) {
            unsigned of2[EG];
#pragma unroll
            for (int o = 0; o < EG; ++o) of2[o] = off[o0 + o];
            row_epilogue<EG, true>(d, *(const f32x16(*)[EG])&acc[o0], of2, (1u << EG) - 1, cl, ctab, C, bnb, s1, s2);
          }
        } else if constexpr (BNB) {
          row_epi_bnb_math<4, BITS, RES>(d, acc, off, 0xfu, cl, ctab, C, eo, s1, s2);
        } else {
          row_epi_res_math<4>(d, acc, off, 0xfu, eo);
        }
        HRP_ASTAMP(4);
        if (d.stats) { const float2 r = bw_reduce16_dpp(s1, s2, lane); vtot0 += r.x; vtot1 += r.y; }
        HRP_ASTAMP(5);
        if (pro != 0 && s + 1 < s_hi) prologue(s + 1, obuf, x2a, bita, x2b, bitb);
        HRP_ASTAMP(6);
        __syncthreads();                                                    // B1: strip s + 1 is ready
        HRP_ASTAMP(7);
      }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if constexpr (FORM == 1) run(T_{}, T_{}, F_{}, F_{});
    else if constexpr (FORM == 2) run(T_{}, F_{}, F_{}, T_{});
    else if constexpr (FORM == 3) run(T_{}, T_{}, T_{}, T_{});
    else run(F_{}, F_{}, F_{}, F_{});
    if (d.stats && any) *(float2*)(stat_lds + (rw * 64 + lane) * 2) = make_float2(vtot0, vtot1);
  }

  // ---- both roles: statistics of the segment (conv_row.h) and, C = 32, the cross-wave combine of the weight gradient
  __syncthreads();
  if (d.stats && any && tid < 2 * C) {
    // channel c, sum `which`: lane (which * 8 + (c & 15) / 2) of both 16-lane rows of half (c >> 4) & 1, element c & 1, of the
    // role-0 waves that own the channel's 32-channel tile
    const int which = tid / C, c = tid - which * C;
    const int mc = c >> 5, hq = (c >> 4) & 1, j = c & 15;
    auto tot = [&](int wh) {
      const int li = (hq * 32 + wh * 8 + (j >> 1)) * 2 + (j & 1);
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (MT == 1 || (w & 1) == mc) t += stat_lds[w * 128 + li] + stat_lds[w * 128 + 32 + li];
      return t;
    };
    float t = tot(which);
    if (bnb && which == 1) t = fmaf(ctab[6 * C + c], t, ctab[7 * C + c] * tot(0));     // sum g * xhat = a * sum g x + b * sum g
    atomicAdd(d.stats + stat_slot * 2 * C + which * C + c, (double)t);
  }
  if (C == 32 && any) {
    const float* dump = (const float*)smem;
    float* ws = (float*)q.workspace + (size_t)slab * (9 * 1024);
    for (int f = tid; f < 9 * 256; f += 512) {       // fixed order: wave 0 + 1 + 2 + 3
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

// Work split of a launch.  A strip of a 64-channel problem costs more than one of a 32-channel problem (twice the MFMAs for the
// same bytes: measured 12.2 against 8.9 us), so workgroups get equal shares of COST, not of strips: problem i occupies the cost
// interval [cost0[i], cost0[i] + nstrips[i] * cost[i]), workgroup w the interval [w T / nwg, (w + 1) T / nwg), and strip k of a
// problem belongs to the workgroup whose interval holds the strip's first cost unit.
struct RowBwArgs {
  hrp_rowbw_desc q[HRP_ROWBW_MAX];
  RowPlan rp[HRP_ROWBW_MAX];
  int cost0[HRP_ROWBW_MAX + 1];
  int cost[HRP_ROWBW_MAX], nstrips[HRP_ROWBW_MAX];
  int first_wg[HRP_ROWBW_MAX];
  int n, total, nwg, pad;        // total: cost units of the launch
};

// strips [k_lo, k_hi) of a problem (cost interval starting at c0, cost c per strip, n strips) that workgroup interval [lo, hi) owns
__host__ __device__ inline void rowbw_range(int lo, int hi, int c0, int c, int n, int& k_lo, int& k_hi) {
  const int a = lo - c0, b = hi - c0;
  k_lo = a <= 0 ? 0 : (a + c - 1) / c;
  k_hi = b <= 0 ? 0 : (b + c - 1) / c;
  if (k_lo > n) k_lo = n;
  if (k_hi > n) k_hi = n;
}

// Problems are addressed with COMPILE-TIME indices into the by-value kernel argument: descriptor fields then are scalar loads
// from the kernarg segment that the compiler re-issues instead of keeping them live (with a run-time problem index, or a table in
// global memory, it kept - and spilled - whole descriptors: 5 000 spilled SGPRs, 1.4 KiB of scratch per lane).
// C0 / C1: channel counts of problem 0 / 1 (C1 == 0: one problem).
template <int C0, int C1, int FORM>
__global__ __launch_bounds__(512) void rowbw_kernel(const RowBwArgs A) {
  const int w = blockIdx.x;
  const int lo = (int)((long long)w * A.total / A.nwg), hi = (int)((long long)(w + 1) * A.total / A.nwg);
  int k0, k1;
  rowbw_range(lo, hi, A.cost0[0], A.cost[0], A.nstrips[0], k0, k1);
  rowbw_body<C0, FORM>(A.q[0], A.rp[0], k0, k1, w - A.first_wg[0], w & (HRP_STAT_SLOTS - 1));
  if constexpr (C1 != 0) {
    rowbw_range(lo, hi, A.cost0[1], A.cost[1], A.nstrips[1], k0, k1);
    __syncthreads();      // (a workgroup that crosses the boundary: the second segment re-initialises the tiles)
    rowbw_body<C1, FORM>(A.q[1], A.rp[1], k0, k1, w - A.first_wg[1], w & (HRP_STAT_SLOTS - 1));
  }
}

// one translation unit per form (conv_rowbw_f*.hip): the three channel combinations of a launch
template <int FORM>
int rowbw_launch_form(const RowBwArgs& A, int grid, int lds_bytes, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)rowbw_kernel<32, 0, FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)rowbw_kernel<64, 0, FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)rowbw_kernel<32, 64, FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int c0 = A.q[0].conv.Cin, c1 = A.n == 2 ? A.q[1].conv.Cin : 0;
  const dim3 g(grid), b(512);
  if (c0 == 32 && c1 == 64) hipLaunchKernelGGL((rowbw_kernel<32, 64, FORM>), g, b, lds_bytes, s, A);
  else if (c0 == 32 && c1 == 0) hipLaunchKernelGGL((rowbw_kernel<32, 0, FORM>), g, b, lds_bytes, s, A);
  else if (c0 == 64 && c1 == 0) hipLaunchKernelGGL((rowbw_kernel<64, 0, FORM>), g, b, lds_bytes, s, A);
  else { set_error("rowbw launch: channel combination (%d, %d)", c0, c1); return HRP_ERR_ARG; }
  return check_launch("rowbw_kernel");
}

#ifdef HRP_TIMELINE
template <int FORM>
int rowbw_timeline_form(void* dst, int nblocks, int clear) {
  if (dst) (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rowbw_timeline), sizeof(unsigned long long) * 32 * nblocks);
  if (clear) {
    void* p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_rowbw_timeline));
    (void)hipMemset(p, 0, sizeof(unsigned long long) * 1024 * 32);
  }
  return 0;
}
#endif

}  // namespace hrp
