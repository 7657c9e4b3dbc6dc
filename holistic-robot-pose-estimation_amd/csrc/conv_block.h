// Fused inference BasicBlock for the high-resolution branches of HRNet (gfx950, bf16):
//     out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)        (reference HRnet.py:41-57, eval mode: BatchNorm = scale / shift)
// for the 3x3 C -> C blocks with C = 32 @ W = 64 and C = 64 @ W = 32 (an image row is 4 KiB), in ONE launch: x is read once,
// out is written once, the intermediate activation h never leaves LDS.  Eval has no batch-statistics barrier between the two
// convolutions, so nothing forces h through HBM.
//
// Structure (roles instead of phases).  A workgroup of 8 waves walks a band of rows of
// one image from top to bottom in STEPS of 4 rows; the two convolutions run concurrently, two steps apart, on different waves:
//   role 0 (waves 0-3)  conv1 step t: h rows [r0 - 1 + 4t, r0 + 3 + 4t) from the x ring -> relu(acc * sc1 + sh1) -> h ring (LDS);
//                       after its epilogue it issues the direct-to-LDS DMA of the 4 new x rows of step t + 2
//   role 1 (waves 4-7)  conv2 step s = t - 2: out rows [r0 + 4s, r0 + 4s + 4) from the h ring; its accumulators are stored one
//                       iteration later, BEFORE the next MFMA loop (residual from the x ring, relu(acc * sc2 + sh2 + x) -> global):
//                       role 1's epilogue runs under role 0's MFMA loop and the other way round
//   ONE barrier per step.  The x ring holds 24 rows: [4i - 10, 4i + 14) of step i are live (residual of step i - 3 .. DMA target
//   of step i + 2: the rows are requested two steps before conv1 reads them); the h ring 12 rows: [4i - 8, 4i + 4).  No row is
//   computed or read twice inside a band; bands of one image (small batches only) recompute one h row and re-read two x rows at
//   each seam.
//   Each role keeps the 9 x KS weight fragments of ITS convolution in registers for the whole band (72 / 144 VGPRs), a wave
//   owns 2 output rows x 32 pixels x 32 output channels per step (18 KS MFMAs 32x32x16 from 12 KS LDS reads).
//   LDS layout, bank swizzle, DMA piece mapping, MFMA row -> channel permutation: conv_row.h.
#pragma once
#include "conv_row.h"

namespace hrp {

template <int C>
struct BlkCfg {
  static constexpr int W = 2048 / C, P = 2 * C, S = P / 16, KS = C / 16, MT = C / 32, NCOL = W / 32;
  static constexpr int PXP = 1024 / P;
  static constexpr int ROWB = (W + 1) * P;          // row + one shared zero pixel
  static constexpr int LEAD = 2;                    // the DMA of a step's rows is issued LEAD steps before conv1 reads them
  static constexpr int RX = 10 + 10 + 4 * (LEAD - 1); // x ring rows: residual of step i - 3 .. DMA target of step i + LEAD
  static constexpr int RH = 12;                     // h ring rows: conv2 of step i - 2 .. conv1 of step i
  static constexpr int XT_OFF = 0;                  // leading zero pixel + ring
  static constexpr int HT_OFF = (P + RX * ROWB + 255) & ~255;
  static constexpr int JUNK_OFF = HT_OFF + ((P + RH * ROWB + 255) & ~255);    // [4 waves] 1 KiB: DMA target of rows outside the image
  static constexpr int CTAB_OFF = JUNK_OFF + 4096;    // [4][C] floats: sc1, sh1, sc2, sh2
  static constexpr int LDS_BYTES = CTAB_OFF + 4 * C * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(MT * NCOL == 2, "two waves side by side, two on top of each other");
  __device__ static __forceinline__ int g(int x) { return C == 32 ? (x >> 2) & 3 : (x >> 1) & 7; }
};

// Development builds (make timeline): s_memtime stamps of iteration 6 of every workgroup, wave 0 of each role
#ifdef HRP_TIMELINE
static __device__ unsigned long long g_block_timeline[256 * 2 * 8];
#define HRP_KSTAMP(k) do { if (lane == 0 && w4 == 0 && blockIdx.x < 256 && i == 6) g_block_timeline[(blockIdx.x * 2 + role) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HRP_KSTAMP(k) do { } while (0)
#endif

struct BlkProblem {
  const void* x;
  void* y;
  const void* w1;
  const void* w2;
  const float *sc1, *sh1, *sc2, *sh2;
  int N, H, bands, band_rows;       // band_rows % 4 == 0, bands * band_rows == H
  int w1_ntaps, w2_ntaps;
  int wslot1[9], wslot2[9];         // packed tap slot of the canonical tap (dy + 1) * 3 + (dx + 1)
  FastDiv fd_bands;
};

struct BlkArgs {
  BlkProblem q[HRP_BLOCK_MAX];
  int first_wg[HRP_BLOCK_MAX];
  int C[HRP_BLOCK_MAX];
  int n, nwg;
};

template <int N>
__device__ __forceinline__ void waitcnt_vm() {
  static_assert(N == 0 || N == 4 || N == 8 || N == 12, "vmcnt immediates used here");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

template <int C>
__device__ __forceinline__ void blk_body(const BlkProblem& q, const int wg) {
  using R = BlkCfg<C>;
  constexpr int W = R::W, P = R::P, S = R::S, KS = R::KS, ROWB = R::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ctab = (float*)(smem + R::CTAB_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, w4 = wave & 3;
  const int l31 = lane & 31, half = lane >> 5;
  const int sel = w4 & 1, rg = w4 >> 1;
  const int col = R::NCOL == 2 ? sel : 0;     // which 32-pixel column block of the rows
  const int m = R::MT == 2 ? sel : 0;         // which 32-channel output tile

  const int n = fdiv(wg, q.fd_bands), band = wg - n * q.bands;
  const int H = q.H, r0 = band * q.band_rows;
  const int nB = q.band_rows >> 2, nA = nB + 1, niter = nB + 3;
  const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);

  // ---- this role's weights: A fragments of the wave's 32 output channels (conv_row.h: MFMA row rho = 8 q + 4 h + i carries
  // output channel 16 h + 4 q + i, so that a lane's 16 accumulators are 16 consecutive channels of one pixel)
  bf16x8 wf[9][KS];
  {
    const int co_lane = m * 32 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    const char* wl = (const char*)(role == 0 ? q.w1 : q.w2) + co_lane * ROW + half * 16;
    const int wnt = role == 0 ? q.w1_ntaps : q.w2_ntaps;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ws = role == 0 ? q.wslot1[t] : q.wslot2[t];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) wf[t][kk] = *(const bf16x8*)(wl + (size_t)((kk * wnt + ws) * C) * ROW);
    }
  }

  // ---- staging (role 0): piece w4 of every row; the lane's 16 bytes = (pixel lane / S of the piece, slot lane % S), holding the
  // LOGICAL slot (lane % S) ^ g(x)
  const int px_in_piece = lane / S;
  const int xcol = w4 * R::PXP + px_in_piece;
  const int lslot = (lane % S) ^ R::g(xcol);
  const char* xg = (const char*)q.x + img_off + (unsigned)(w4 * 1024 + px_in_piece * P + lslot * 16);
  char* xrows = smem + R::XT_OFF + P;
  char* hrows = smem + R::HT_OFF + P;
  // `cnt` rows from ring index qi (image row r0 - 2 + qi) into ring slots slot, slot + 1 .. (mod RX).  Every row costs every lane
  // exactly ONE DMA operation - a row outside the image is zero-filled and its DMA goes to a junk piece - so that
  // "all but the last 4 (LEAD - 1) operations have landed" is a constant vmcnt
  auto stage = [&](const int qi, int slot, const int cnt) {
    for (int k = 0; k < cnt; ++k) {
      const int r = r0 - 2 + qi + k;
      char* dst = xrows + slot * ROWB + w4 * 1024;
      if (r >= 0 && r < H) dma16(xg + r * (W * P), dst);
      else {
        *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
        dma16(xg, smem + R::JUNK_OFF + w4 * 1024);
      }
      slot = slot + 1 == R::RX ? 0 : slot + 1;
    }
  };
  if (role == 0) stage(0, 0, 6 + 4 * (R::LEAD - 1));
  // the zero pixels of both tiles: one in front of ring row 0, one behind every ring row
  if (tid < (R::RX + 1 + R::RH + 1) * S) {
    const int tile = tid >= (R::RX + 1) * S, e = tid - tile * ((R::RX + 1) * S);
    const int k = e / S, j = e - k * S;
    *(uint4*)(smem + (tile ? R::HT_OFF : R::XT_OFF) + (k == 0 ? 0 : P + (k - 1) * ROWB + W * P) + j * 16) = make_uint4(0, 0, 0, 0);
  }
  if (tid >= 256 && tid < 256 + C) {
    const int c = tid - 256;
    ctab[0 * C + c] = q.sc1[c]; ctab[1 * C + c] = q.sh1[c];
    ctab[2 * C + c] = q.sc2[c]; ctab[3 * C + c] = q.sh2[c];
  }

  // read address of (dx, kk = 0) in ring row 0: pixel x = col*32 + l31 + dx (x = -1 / W are the shared zero pixels); the K chunk
  // kk is an XOR of bits 5.. (slot' = (2 kk + half) ^ g(x))
  int a0[3];
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi) {
    const int xq = col * 32 + l31 + dxi - 1;
    a0[dxi] = P + xq * P + ((half ^ R::g(xq)) << 4);
  }
  const int xpix = col * 32 + l31;
  const int cl = m * 32 + 16 * half;                         // first output channel of the lane
  // the lane's two 16-byte slots of its pixel (channels cl .. cl + 15) inside a ring row
  int eo[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) eo[hh] = xpix * P + (((m * 4 + 2 * half + hh) ^ R::g(xpix)) << 4);
  char* yg = (char*)q.y + img_off + (unsigned)(xpix * P + cl * 2);

  if (role == 0) waitcnt_vm<4 * (R::LEAD - 1)>();             // the six rows of step 0
  __syncthreads();

  // ring slots of the rows with index 4 i (x: of image row r0 - 2 + 4 i, h: of image row r0 - 1 + 4 i), kept incrementally
  int xb = 0, hb = 0;
  auto wrap = [](const int v, const int n) { return v >= n ? v - n : v; };
  f32x16 acc[2];                                             // (role 1: alive across the barrier, stored by the NEXT iteration)
  // the lane's 16 + 16 per-channel constants of this role's BatchNorm: resident for the band where the registers allow it (C = 32;
  // the 64-channel kernel holds 144 registers of weights and re-reads the 8 + 8 of a half from LDS)
  constexpr bool RESIDENT = C == 32;
  const float* cp = ctab + (role == 0 ? 0 : 2 * C) + cl;
  float csc[RESIDENT ? 16 : 1], csh[RESIDENT ? 16 : 1];
  if constexpr (RESIDENT) {
#pragma unroll
    for (int e = 0; e < 16; e += 4) {
      const float4 a = *(const float4*)(cp + e), b = *(const float4*)(cp + C + e);
      csc[e] = a.x; csc[e + 1] = a.y; csc[e + 2] = a.z; csc[e + 3] = a.w;
      csh[e] = b.x; csh[e + 1] = b.y; csh[e + 2] = b.z; csh[e + 3] = b.w;
    }
  }
  auto consts = [&](const int hh, float (&sc)[8], float (&sh)[8]) {
    if constexpr (RESIDENT) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = csc[8 * hh + e]; sh[e] = csh[8 * hh + e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const float4 a = *(const float4*)(cp + 8 * hh + e), b = *(const float4*)(cp + C + 8 * hh + e);
        sc[e] = a.x; sc[e + 1] = a.y; sc[e + 2] = a.z; sc[e + 3] = a.w;
        sh[e] = b.x; sh[e + 1] = b.y; sh[e + 2] = b.z; sh[e + 3] = b.w;
      }
    }
  };
  uint4 rres[2][2];                                          // role 1: the residual of the accumulated step, read one iteration early
  for (int i = 0; i < niter; ++i) {
    HRP_KSTAMP(0);
    if (role == 1 && i >= 3) {
      // ---- role 1 first stores the step it accumulated in the previous iteration (conv2 step i - 3): its VALU / store work
      // runs while role 0 is in its MFMA loop, and its own MFMA loop below while role 0 is in ITS epilogue
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float sc[8], sh[8];
        consts(hh, sc, sh);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int yr = 4 * (i - 3) + 2 * rg + o;           // row of the band
          float v[8], r[8];
          Elem<bf16_t>::unpack(rres[o][hh], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(acc[o][8 * hh + e] * sc[e] + sh[e] + r[e], 0.f);
          *(uint4*)(yg + (unsigned)((r0 + yr) * (W * P)) + 16 * hh) = Elem<bf16_t>::pack(v);
        }
      }
    }
    HRP_KSTAMP(1);
    const bool act = role == 0 ? i < nA : (i >= 2 && i - 2 < nB);
    if (act) {
      // the wave's first input row: conv1 x index 4 i + 2 rg, conv2 h index 4 (i - 2) + 2 rg
      const int s0 = role == 0 ? xb + 2 * rg : wrap(hb + R::RH - 8 + 2 * rg, R::RH);
      const char* tile = smem + (role == 0 ? R::XT_OFF : R::HT_OFF);
      int roff[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) roff[k] = wrap(s0 + k, role == 0 ? R::RX : R::RH) * ROWB;
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
      {
        constexpr int NSTEP = 4 * 3 * KS, RINGB = 4, AHEAD = 3;
        bf16x8 bq[RINGB];
        auto rd = [&](int s) -> bf16x8 {   // s is a constant after unrolling
          const int irel = s / (3 * KS), dxi = (s / KS) % 3, kk = s % KS;
          return *(const bf16x8*)(tile + roff[irel] + (a0[dxi] ^ (kk << 5)));
        };
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) bq[s % RINGB] = rd(s);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
          if (s + AHEAD < NSTEP) bq[(s + AHEAD) % RINGB] = rd(s + AHEAD);
          const int irel = s / (3 * KS), dxi = (s / KS) % 3, kk = s % KS;
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            const int dyi = irel - o;               // input row p0 + irel = output row (2 rg + o) + dy, dy = dyi - 1
            if (dyi >= 0 && dyi <= 2)
              acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[dyi * 3 + dxi][kk], bq[s % RINGB], acc[o], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);        // keep the read-ahead where it is
        }
      }
      HRP_KSTAMP(2);
      if (role == 0) {
        // ---- conv1 epilogue: lane = pixel xpix of the wave's rows, channels cl .. cl + 15 -> h ring
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          float sc[8], sh[8];
          consts(hh, sc, sh);
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            const int hr = r0 - 1 + 4 * i + 2 * rg + o;      // image row of the h row
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(acc[o][8 * hh + e] * sc[e] + sh[e], 0.f);
            uint4 pk = Elem<bf16_t>::pack(v);
            if (hr < 0 || hr >= H) pk = make_uint4(0, 0, 0, 0);      // (rows outside the image are conv2's zero padding; uniform)
            *(uint4*)(hrows + wrap(hb + 2 * rg + o, R::RH) * ROWB + eo[hh]) = pk;
          }
        }
      } else {
        // the residual of the step just accumulated (x rows of conv2 step i - 2: ring index 4 (i - 2) + 2 rg + o + 2), for the store
        // at the start of the next iteration
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) rres[o][hh] = *(const uint4*)(xrows + wrap(xb + R::RX - 6 + 2 * rg + o, R::RX) * ROWB + eo[hh]);
      }
    }
    HRP_KSTAMP(3);
    if (role == 0) {
      // the rows of step i + LEAD: requested AFTER this role's MFMA loop and epilogue, so that the issue cost of the four pieces
      // (~180 cycles each next to a partner in its MFMA loop) lies beside role 1's matrix phase, not in front of this role's own
      if (i + R::LEAD < nA) stage(4 * i + 6 + 4 * (R::LEAD - 1), wrap(xb + 6 + 4 * (R::LEAD - 1), R::RX), 4);
      else
        for (int k = 0; k < 4; ++k) dma16(xg, smem + R::JUNK_OFF + w4 * 1024);     // (keeps the operation count per step)
      waitcnt_vm<4 * (R::LEAD - 1)>();                        // the rows of step i + 1 have landed
    }
    HRP_KSTAMP(4);
    __syncthreads();
    HRP_KSTAMP(5);
    xb = wrap(xb + 4, R::RX);
    hb = wrap(hb + 4, R::RH);
  }
}

// Problems are addressed with compile-time indices into the by-value kernel argument.  C0 / C1: channel counts
// of problem 0 / 1 (C1 == 0: one problem).
template <int C0, int C1>
__global__ __launch_bounds__(512) void block_kernel(const BlkArgs A) {
  const int w = blockIdx.x;
  if constexpr (C1 == 0) {
    blk_body<C0>(A.q[0], w);
  } else {
    if (w < A.first_wg[1]) blk_body<C0>(A.q[0], w);
    else blk_body<C1>(A.q[1], w - A.first_wg[1]);
  }
}

}  // namespace hrp
