// Convolution weight gradient on MFMA for gfx950.
//
//   dW[co][ci][t] (+)= sum over output pixels p of dY[p][co] * X[p shifted by tap t][ci]
//
// GEMM view per tap: M = cout, N = cin, K = pixels (the long dimension).  A workgroup owns one
// 32-cout x 32-cin block for ALL taps and walks a strided share of the pixel tiles; per tile it stages
// the dY tile and the X halo tile (32 channels each) in LDS once and every tap re-reads the X tile at a
// shifted offset.  The 4 waves split the tile's pixels (split-K inside the workgroup), partial sums are
// combined through LDS and leave as one partial slab per workgroup (folded into the PyTorch-shaped gradient
// by wgrad_reduce_kernel) or, without a workspace, as fp32 atomics.
//
// Operand gather: the reduction index is the pixel, but NHWC keeps channels contiguous, so the 8 k-values
// a lane needs for the bf16 MFMA belong to 8 different pixels.  gfx950's LDS transpose read
// (ds_read_b64_tr_b16) does exactly that re-layout: within a 16-lane group source lane 4j+t supplies 4
// contiguous channels (8 bytes) of pixel j and destination lane i receives channel i of pixels 0..3
// (mapping measured with tools/probe_tr.hip: result[i][j] = src[4j + (i >> 2)][i & 3]).  Two such reads
// build one 8-deep MFMA operand; every source lane carries its own pixel address, so tap shifts and
// stride-2 sampling cost nothing extra.  fp32 needs one 32-bit read per operand.
#include "hrp_common.h"
#include "batch.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// -DHRP_TIMELINE (tools/bench_kernels.py timeline, never in the shipped library): wave 0 of every workgroup
// stamps the 100 MHz wall clock at phase boundaries into the last MiB of the workspace.
#ifdef HRP_TIMELINE
#define HRP_STAMP(i) do { if (tid == 0) tl[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HRP_STAMP(i) do { } while (0)
#endif

#ifndef WGRAD_OCC
#define WGRAD_OCC
#endif

namespace hrp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int WSTAGE_U = 8;

struct WgradTiling {
  int TH, TW, TI, BM;
  int IHt, IWt, mindy, mindx;
  int tiles_x, tiles_y, tiles_n, ntiles;
  int n_cob, n_cib, G;
  int in_pix;
  int x_pieces, dy_pieces, buf_bytes;   // 1 KiB DMA pieces of the X halo tile / dY tile; one stage buffer
  int lds_tab_off, lds_red_off;
  int use_ws, lds_bytes;
  // batched launch: workgroup -> (pixel share, channel-block pair) so that the workgroups which read the same 32-channel
  // slices of x / dY sit on ONE XCD (its L2 serves the re-reads).  xmode 1: >= 8 pairs - XCD k owns an xa x xb block of
  // (cout, cin) blocks for every pixel share; xmode 2: 1 / 2 / 4 pairs - XCD k owns whole pixel shares; 0: pair-major.
  int xmode, xa, xb, xnb;
  FastDiv fd_ihw, fd_iwt, fd_thw, fd_tw, fd_tx, fd_ty, fd_cib, fd_g;
};

// Workgroup ids go round-robin over the 8 XCDs; a weight-gradient workgroup reads the cin slice `cib` of x and the cout
// slice `cob` of dY for its pixel share, so with pair-major numbering every slice is fetched from HBM by n_cob (n_cib)
// different L2s: 28.4 GB per step against 18.5 GB of operands (profiles/r03_traffic.json).
__device__ __forceinline__ void wgrad_block_of(const WgradTiling& t, const int local, int& gxi, int& blk) {
  if (t.xmode == 1) {
    const int j = local >> 3, xk = local & 7, ppx = t.xa * t.xb;
    gxi = j / ppx;
    const int r = j - gxi * ppx, ka = xk / t.xnb, kb = xk - ka * t.xnb;
    blk = (ka * t.xa + r / t.xb) * t.n_cib + kb * t.xb + r % t.xb;
  } else if (t.xmode == 2) {
    const int j = local >> 3, xk = local & 7, pairs = t.n_cob * t.n_cib, q = j / pairs;
    gxi = q * 8 + xk;
    blk = j - q * pairs;
  } else {
    blk = fdiv(local, t.fd_g);        // consecutive workgroups = the pixel shares of one (cout, cin) pair
    gxi = local - blk * t.G;
  }
}

template <typename T>
static constexpr int wgrad_maxp_x() { return std::is_same<T, f32x3_t>::value ? 10 : 8; }

template <typename T>
struct WG;
template <>
struct WG<bf16_t> {
  static constexpr int K = 16;
  using Frag = bf16x8;
  __device__ static __forceinline__ bf16x4 tr(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
  }
  __device__ static __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct WG<float> {
  static constexpr int K = 2;
  using Frag = float;
  __device__ static __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

__device__ uint4 g_wg_zero_page[4];  // 64 zero bytes: DMA source of padding / out-of-range rows

// 64 lanes x 16 bytes global -> LDS (lane-linear at lds_wave_base).  Written as inline assembly on purpose:
// behind the builtin the compiler's wait-count pass cannot tell the DMA destination from the buffer the
// transpose reads are working on and drains the DMA (s_waitcnt vmcnt(0)) before every LDS read, which
// serialises the prefetch with the MFMAs.  The kernel waits explicitly (vmcnt(0) + barrier) per tile.
__device__ __forceinline__ void wg_dma16(const char* src, char* lds_wave_base) {
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"(src), "s"(lds) : "memory");   // m0 is scratch for the compiler too: it never keeps a value there
}

// ---- HRP_F32X3: fp32 tensors, every product as three bf16 MFMAs on split operands (hi = rne(v), lo = rne(v - hi)) -----------
// The fp32 tile of a workgroup is staged by DMA into a RAW region, converted ONCE into bf16 hi / lo planes laid out like the bf16
// program's tiles (64-byte pixel rows), and the wave program is the bf16 one - transpose reads, operand offsets in registers - with
// a hi and a lo fragment per operand and lo*hi + hi*lo + hi*hi per tap.  (The first version gathered eight ds_read_b32 per operand
// and split in registers per tap: ~350 instructions per 27 MFMAs, 24.1 ms of weight gradients per step; this one ~70.)
// LDS: RAW (DMA target) + the planes, each one tile - the DMA of tile i + 1 runs under the MFMAs of tile i because RAW is free
// again once tile i has been converted.
template <int NT, int NKS>
__device__ __forceinline__ void conv_wgrad_x3_body(const hrp_wgrad_desc& d, const WgradTiling& t, const int gxi, const int blk) {
  constexpr int SZ = 4, VEC = 4, P = 128, NVEC = 8;      // the RAW tile: fp32, 32 channels per pixel row
  constexpr int P2 = 64;                                   // a plane's pixel row: 32 bf16
  constexpr int MAXP_X = wgrad_maxp_x<f32x3_t>(), MAXP_DY = 6;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int npairs = t.n_cob * t.n_cib;
  const int cob = fdiv(blk, t.fd_cib), cib = blk - cob * t.n_cib;
  const int co0 = cob * 32, ci0 = cib * 32;
  const int IS = d.in_stride;
  const int thw = t.TH * t.TW, ihw = t.IHt * t.IWt;

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const char* xg = (const char*)d.x;
  const char* dyg = (const char*)d.dy;
  const char* zero = (const char*)g_wg_zero_page;
  asm volatile("" : "+v"(zero));
  const int ppw = t.BM / 4;  // pixels per wave

  int tapoff[NT];
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) tapoff[tp] = ((d.dy_t[tp] - t.mindy) * t.IWt + (d.dx_t[tp] - t.mindx)) * P2;

  const int x_pieces = t.x_pieces, dy_pieces = t.dy_pieces;
  int xrel[MAXP_X], xcode[MAXP_X], dyrel[MAXP_DY], dycode[MAXP_DY];
  {
    const int y_last = (t.tiles_y - 1) * t.TH, x_last = (t.tiles_x - 1) * t.TW, n_last = (t.tiles_n - 1) * t.TI;
    const int iy_last = y_last * IS + t.mindy, ix_last = x_last * IS + t.mindx;
#pragma unroll
    for (int i = 0; i < MAXP_X; ++i) {
      xcode[i] = 32; xrel[i] = 0;
      if (wave + 4 * i >= x_pieces) continue;
      const int sl = (wave + 4 * i) * 64 + lane, pix = sl / NVEC, vec = sl - pix * NVEC;
      const int ti = fdiv16(pix, t.fd_ihw), rem = pix - mul24(ti, ihw);
      const int iy = fdiv16(rem, t.fd_iwt), ix = rem - mul24(iy, t.IWt);
      const int c = ci0 + vec * VEC;
      int code = (pix >= t.in_pix || c >= d.Cin) ? 32 : 0;
      code |= (iy + t.mindy < 0) ? 1 : 0;
      code |= (iy + iy_last >= d.H) ? 2 : 0;
      code |= (ix + t.mindx < 0) ? 4 : 0;
      code |= (ix + ix_last >= d.W) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      xcode[i] = code;
      xrel[i] = mul24(mul24(mul24(ti, d.H) + iy, d.W) + ix, d.x_pitch * SZ) + c * SZ;
    }
#pragma unroll
    for (int i = 0; i < MAXP_DY; ++i) {
      dycode[i] = 32; dyrel[i] = 0;
      if (wave + 4 * i >= dy_pieces) continue;
      const int sl = (wave + 4 * i) * 64 + lane, m = sl / NVEC, vec = sl - m * NVEC;
      const int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      const int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      const int c = co0 + vec * VEC;
      int code = (m >= t.BM || ti >= t.TI || c >= d.Cout) ? 32 : 0;
      code |= (ty + y_last >= d.Ho) ? 2 : 0;
      code |= (tx + x_last >= d.Wo) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      dycode[i] = code;
      dyrel[i] = mul24(mul24(mul24(ti, d.Ho) + ty, d.Wo) + tx, d.dy_pitch * SZ) + c * SZ;
    }
  }

  struct TileCtx { const char* xbase; const char* dybase; int cls; };
  auto tile_ctx = [&](int tile) {
    int q = fdiv(tile, t.fd_tx);
    const int tx_i = tile - q * t.tiles_x;
    const int tn_i = fdiv(q, t.fd_ty);
    const int ty_i = q - tn_i * t.tiles_y;
    const int n0 = tn_i * t.TI, oy0 = ty_i * t.TH, ox0 = tx_i * t.TW;
    const int iy0 = oy0 * IS + t.mindy, ix0 = ox0 * IS + t.mindx;
    TileCtx c;
    c.cls = 32 | (ty_i == 0 ? 1 : 0) | (ty_i == t.tiles_y - 1 ? 2 : 0) | (tx_i == 0 ? 4 : 0) |
            (tx_i == t.tiles_x - 1 ? 8 : 0) | (tn_i == t.tiles_n - 1 ? 16 : 0);
    c.xbase = xg + (((long long)n0 * d.H + iy0) * d.W + ix0) * (long long)d.x_pitch * SZ;
    c.dybase = dyg + (((long long)n0 * d.Ho + oy0) * d.Wo + ox0) * (long long)d.dy_pitch * SZ;
    return c;
  };
  char* const raw = smem;                         // [x_pieces + dy_pieces] KiB, the tile as it lies in HBM
  char* const pl = smem + t.buf_bytes;            // X hi | X lo | dY hi | dY lo
  const int xpl = x_pieces * 512, dypl = dy_pieces * 512;
  auto issue_slot = [&](const TileCtx& c, int slot) {   // slot is a constant after unrolling
    if (slot < MAXP_X) {
      const int p = wave + 4 * slot;
      if (p < x_pieces) wg_dma16((xcode[slot] & c.cls) ? zero : c.xbase + (unsigned)xrel[slot], raw + p * 1024);
    } else if (slot < MAXP_X + MAXP_DY) {
      const int i = slot - MAXP_X, p = wave + 4 * i;
      if (p < dy_pieces) wg_dma16((dycode[i] & c.cls) ? zero : c.dybase + (unsigned)dyrel[i], raw + (x_pieces + p) * 1024);
    }
  };

  const int tr_pix = (lane & 15) >> 2;
  const int tr_coff = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  if (gxi < t.ntiles) {
    const TileCtx c = tile_ctx(gxi);
#pragma unroll
    for (int slot = 0; slot < MAXP_X + MAXP_DY; ++slot) issue_slot(c, slot);
  }
  int xo0[NKS], xo1[NKS], ao[NKS];
  {
    auto xoff = [&](int m) {
      int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      if (ti >= t.TI) ti = t.TI - 1;
      return mul24(mul24(mul24(ti, t.IHt) + mul24(ty, IS), t.IWt) + mul24(tx, IS), P2) + tr_coff;
    };
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int m0 = wave * ppw + ks * 16 + 8 * khalf + tr_pix;
      xo0[ks] = xoff(m0); xo1[ks] = xoff(m0 + 4);
      ao[ks] = m0 * P2 + tr_coff;
    }
  }
  // conversion: one 32-byte granule (8 floats) of RAW -> 16 bytes of the hi plane + 16 bytes of the lo plane
  const int ngran_x = t.in_pix * 4, ngran = ngran_x + t.BM * 4;

  for (int tile = gxi; tile < t.ntiles; tile += t.G) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // the tile has landed in RAW; every wave is done with the planes of the previous tile
    for (int e = tid; e < ngran; e += 256) {
      const bool isx = e < ngran_x;
      const int r = isx ? e : e - ngran_x;
      const char* src = raw + (isx ? 0 : x_pieces * 1024) + r * 32;
      const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 16);
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      uint4 hi, lo;
      split_bf16x8(x, hi, lo);
      char* dst = pl + (isx ? 0 : 2 * xpl) + r * 16;
      *(uint4*)dst = hi;
      *(uint4*)(dst + (isx ? xpl : dypl)) = lo;
    }
    __syncthreads();
    const bool more = tile + t.G < t.ntiles;
    TileCtx nx{};
    if (more) nx = tile_ctx(tile + t.G);
    const char* lds_x = pl;
    const char* lds_dy = pl + 2 * xpl;
    constexpr int D = NT >= 2 ? 2 : NT, TOT = NKS * NT;
    bf16x8 ah[2], al[2], bh[D + 1], bl[D + 1];
    auto tr2 = [&](const char* p0, const char* p1) {
      bf16x4 lo = WG<bf16_t>::tr(p0), hi = WG<bf16_t>::tr(p1);
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_a = [&](int ks, bf16x8& fh, bf16x8& fl) {
      fh = tr2(lds_dy + ao[ks], lds_dy + ao[ks] + 4 * P2);
      fl = tr2(lds_dy + dypl + ao[ks], lds_dy + dypl + ao[ks] + 4 * P2);
    };
    auto load_b = [&](int q, bf16x8& fh, bf16x8& fl) {
      const int ks = q / NT, tp = q % NT;   // constants after unrolling
      fh = tr2(lds_x + xo0[ks] + tapoff[tp], lds_x + xo1[ks] + tapoff[tp]);
      fl = tr2(lds_x + xpl + xo0[ks] + tapoff[tp], lds_x + xpl + xo1[ks] + tapoff[tp]);
    };
    // RAW is free: the next tile's DMA goes out in one burst, so that all of it has the whole MFMA phase to land (pieces issued
    // between the MFMAs as in the bf16 program: 23.8 instead of 22.8 ms of weight gradients per step - a tile is 54 MFMAs here)
    if (more) {
#pragma unroll
      for (int slot = 0; slot < MAXP_X + MAXP_DY; ++slot) issue_slot(nx, slot);
    }
    load_a(0, ah[0], al[0]);
#pragma unroll
    for (int q = 0; q < D && q < TOT; ++q) load_b(q, bh[q % (D + 1)], bl[q % (D + 1)]);
#pragma unroll
    for (int q = 0; q < TOT; ++q) {
      const int ks = q / NT, tp = q % NT, r = q % (D + 1);
      if (tp == 0 && ks + 1 < NKS) load_a(ks + 1, ah[(ks + 1) & 1], al[(ks + 1) & 1]);
      if (q + D < TOT) load_b(q + D, bh[(q + D) % (D + 1)], bl[(q + D) % (D + 1)]);
      acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks & 1], bh[r], acc[tp], 0, 0, 0);      // (small terms first)
      acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks & 1], bl[r], acc[tp], 0, 0, 0);
      acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks & 1], bh[r], acc[tp], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- combine the 4 waves' partial sums through LDS and write the partial slab (conv_wgrad_body's epilogue) -------------
  __syncthreads();
  float* dump = (float*)smem;
  const int tstride = d.dw_tap_stride > 0 ? d.dw_tap_stride : d.ntaps;
  float* ws = t.use_ws ? (float*)d.workspace + ((size_t)gxi * npairs + blk) * (NT * 1024) : nullptr;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
    float* mine = dump + wave * (NT * 512) + 4 * khalf * 32 + l31;
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
      for (int j = 0; j < 8; ++j) mine[(tp * 16 + (j & 3) + 8 * (j >> 2)) * 32] = acc[tp][8 * h + j];
    __syncthreads();
    for (int f = tid; f < NT * 128; f += 256) {
      float4 v = ((const float4*)dump)[f];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 u = ((const float4*)(dump + w * (NT * 512)))[f];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      const int tp = f >> 7, rem = f & 127;
      if (ws) {
        ((float4*)(ws + tp * 1024 + 512 * h))[rem] = v;
      } else {
        const int co = co0 + 16 * h + (rem >> 3), cin = ci0 + 4 * (rem & 7);
        if (co < d.Cout) {
          float* o = d.dw + ((size_t)co * d.dw_cin + cin) * tstride + d.dw_tap_off + tp;
          if (cin < d.dw_cin) atomicAdd(o, v.x);
          if (cin + 1 < d.dw_cin) atomicAdd(o + tstride, v.y);
          if (cin + 2 < d.dw_cin) atomicAdd(o + 2 * tstride, v.z);
          if (cin + 3 < d.dw_cin) atomicAdd(o + 3 * tstride, v.w);
        }
      }
    }
  }
}

// NB = 32-channel blocks per workgroup in each of the cout / cin dimensions.  NB = 2 (1x1 layers, bf16): a
// 64 x 64 block of dW per workgroup = 4 MFMAs per 4 fragment reads instead of 1 per 2, and half the re-reads of
// X and dY across workgroups (the 32 x 32 version of the 1x1 layers ran at 80 TFLOP/s, LDS-read bound).
// gxi: which of the t.G pixel-tile shares this workgroup walks; blk: its (cout block, cin block) pair.  The single
// launch passes (blockIdx.x, blockIdx.y), the batched launch what it decoded from its linear block index.
template <typename T, int NT, int NKS, int NB>
__device__ __forceinline__ void conv_wgrad_body(const hrp_wgrad_desc& d, const WgradTiling& t, const int gxi, const int blk) {
  static_assert(NB == 1 || (NT == 1 && Elem<T>::SZ == 2), "NB = 2 is built for 1x1 bf16 only");
  if constexpr (std::is_same<T, f32x3_t>::value) {
    conv_wgrad_x3_body<NT, (NKS > 0 ? NKS : 1)>(d, t, gxi, blk);
    return;
  } else {
  constexpr int SZ = Elem<T>::SZ, VEC = Elem<T>::VEC;
  constexpr int CB = 32 * NB;       // channels per workgroup block (cout and cin)
  constexpr int NTE = NT * NB * NB; // accumulator tiles: [tap][cout block][cin block]
  constexpr int P = CB * SZ;        // LDS pixel row: CB channels, unpadded (DMA writes lane-linear)
  constexpr int NVEC = P / 16;      // 16-byte slots per pixel row
  constexpr int K = WG<T>::K;
  // 1 KiB DMA pieces per wave (host keeps tiles below 32 / 24 KiB; fp32x3: 40 KiB - its smallest tile is 64 pixels, whose fp32
  // halo tile under a stride-2 3x3 layer is 37 KiB)
  constexpr int MAXP_X = wgrad_maxp_x<T>(), MAXP_DY = 6;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* xtab = (int*)(smem + t.lds_tab_off);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int npairs = t.n_cob * t.n_cib;
  const int cob = fdiv(blk, t.fd_cib), cib = blk - cob * t.n_cib;
  const int co0 = cob * CB, ci0 = cib * CB;
  const int IS = d.in_stride;
  const int thw = t.TH * t.TW, ihw = t.IHt * t.IWt;
#ifdef HRP_TIMELINE
  unsigned long long* tl = (unsigned long long*)((char*)d.workspace + d.workspace_bytes - (1 << 20)) +
                           ((size_t)gxi * npairs + blk) * 16;
#endif
  HRP_STAMP(0);

  // pixel -> X-tile byte offset table (fp32 path; bf16 keeps its operand offsets in registers)
  if constexpr (SZ == 4) {
    for (int m = tid; m < t.BM; m += 256) {
      int ti = fdiv(m, t.fd_thw), rem = m - ti * thw;
      int ty = fdiv(rem, t.fd_tw), tx = rem - ty * t.TW;
      if (ti >= t.TI) ti = t.TI - 1;  // idle slot (its dY row is zero)
      xtab[m] = ((ti * t.IHt + ty * IS) * t.IWt + tx * IS) * P;
    }
  }

  f32x16 acc[NTE];
#pragma unroll
  for (int i = 0; i < NTE; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const char* xg = (const char*)d.x;
  const char* dyg = (const char*)d.dy;
  const char* zero_sym = (const char*)g_wg_zero_page;
  const char* zero = zero_sym;
  // keep the zero-page pointer in a VGPR pair: left to itself the compiler rematerialises it (s_getpc, 2 s_add,
  // 2 v_mov) in front of every DMA piece, a third of the instructions of the K loop
  asm volatile("" : "+v"(zero));
  const int ppw = t.BM / 4;  // pixels per wave

  int tapoff[NT];
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) tapoff[tp] = ((d.dy_t[tp] - t.mindy) * t.IWt + (d.dx_t[tp] - t.mindx)) * P;

  // ---- DMA plan (tile independent part) ---------------------------------------------------------------
  // Per 1 KiB piece a lane keeps the byte offset of its 16-byte slot relative to the tile origin and a
  // validity code: bit 0/1 = outside the image in the first / last tile row, bit 2/3 = same for tile
  // columns, bit 4 = beyond the batch in the last image group, bit 5 = never fetched (tile padding,
  // channel tail).  A tile's class mask selects the bits that apply to it, so issuing a piece is one AND,
  // one select and one 64-bit add (the host checks that only border tiles can leave the image).
  const int x_pieces = t.x_pieces, dy_pieces = t.dy_pieces;
  int xrel[MAXP_X], xcode[MAXP_X], dyrel[MAXP_DY], dycode[MAXP_DY];
  {
    const int y_last = (t.tiles_y - 1) * t.TH, x_last = (t.tiles_x - 1) * t.TW, n_last = (t.tiles_n - 1) * t.TI;
    const int iy_last = y_last * IS + t.mindy, ix_last = x_last * IS + t.mindx;
#pragma unroll
    for (int i = 0; i < MAXP_X; ++i) {
      xcode[i] = 32; xrel[i] = 0;
      if (wave + 4 * i >= x_pieces) continue;
      const int sl = (wave + 4 * i) * 64 + lane, pix = sl / NVEC, vec = sl - pix * NVEC;
      const int ti = fdiv16(pix, t.fd_ihw), rem = pix - mul24(ti, ihw);
      const int iy = fdiv16(rem, t.fd_iwt), ix = rem - mul24(iy, t.IWt);
      const int c = ci0 + vec * VEC;
      int code = (pix >= t.in_pix || c >= d.Cin) ? 32 : 0;
      code |= (iy + t.mindy < 0) ? 1 : 0;
      code |= (iy + iy_last >= d.H) ? 2 : 0;
      code |= (ix + t.mindx < 0) ? 4 : 0;
      code |= (ix + ix_last >= d.W) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      xcode[i] = code;
      xrel[i] = mul24(mul24(mul24(ti, d.H) + iy, d.W) + ix, d.x_pitch * SZ) + c * SZ;
    }
#pragma unroll
    for (int i = 0; i < MAXP_DY; ++i) {
      dycode[i] = 32; dyrel[i] = 0;
      if (wave + 4 * i >= dy_pieces) continue;
      const int sl = (wave + 4 * i) * 64 + lane, m = sl / NVEC, vec = sl - m * NVEC;
      const int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      const int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      const int c = co0 + vec * VEC;
      int code = (m >= t.BM || ti >= t.TI || c >= d.Cout) ? 32 : 0;
      code |= (ty + y_last >= d.Ho) ? 2 : 0;
      code |= (tx + x_last >= d.Wo) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      dycode[i] = code;
      dyrel[i] = mul24(mul24(mul24(ti, d.Ho) + ty, d.Wo) + tx, d.dy_pitch * SZ) + c * SZ;
    }
  }

  struct TileCtx { const char* xbase; const char* dybase; char* buf; int cls; };
  auto tile_ctx = [&](int tile, char* buf) {
    int q = fdiv(tile, t.fd_tx);
    const int tx_i = tile - q * t.tiles_x;
    const int tn_i = fdiv(q, t.fd_ty);
    const int ty_i = q - tn_i * t.tiles_y;
    const int n0 = tn_i * t.TI, oy0 = ty_i * t.TH, ox0 = tx_i * t.TW;
    const int iy0 = oy0 * IS + t.mindy, ix0 = ox0 * IS + t.mindx;
    TileCtx c;
    c.cls = 32 | (ty_i == 0 ? 1 : 0) | (ty_i == t.tiles_y - 1 ? 2 : 0) | (tx_i == 0 ? 4 : 0) |
            (tx_i == t.tiles_x - 1 ? 8 : 0) | (tn_i == t.tiles_n - 1 ? 16 : 0);
    // tile origins (the X origin may lie before the tensor: only valid lanes dereference it)
    c.xbase = xg + (((long long)n0 * d.H + iy0) * d.W + ix0) * (long long)d.x_pitch * SZ;
    c.dybase = dyg + (((long long)n0 * d.Ho + oy0) * d.Wo + ox0) * (long long)d.dy_pitch * SZ;
    c.buf = buf;
    return c;
  };
  // DMA slot = one 1 KiB piece of this wave: slots 0 .. MAXP_X-1 belong to the X tile, the rest to dY
  auto issue_slot = [&](const TileCtx& c, int slot) {   // slot is a constant after unrolling
    if (slot < MAXP_X) {
      const int p = wave + 4 * slot;
      if (p < x_pieces) wg_dma16((xcode[slot] & c.cls) ? zero : c.xbase + (unsigned)xrel[slot], c.buf + p * 1024);
    } else if (slot < MAXP_X + MAXP_DY) {
      const int i = slot - MAXP_X, p = wave + 4 * i;
      if (p < dy_pieces)
        wg_dma16((dycode[i] & c.cls) ? zero : c.dybase + (unsigned)dyrel[i], c.buf + (x_pieces + p) * 1024);
    }
  };
  auto issue = [&](int tile, char* buf) {
    const TileCtx c = tile_ctx(tile, buf);
#pragma unroll
    for (int slot = 0; slot < MAXP_X + MAXP_DY; ++slot) issue_slot(c, slot);
  };

  // transpose-read lane roles (bf16): source lane s = lane & 15 -> pixel (s >> 2) of the 4-pixel block,
  // channels rbase + 4 (s & 3) .. +3; rbase = 16 for the odd 16-lane groups
  const int tr_pix = (lane & 15) >> 2;
  const int tr_coff = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * SZ;

  // tile loop, double buffered: the DMA of tile i+1 runs under the MFMAs of tile i.  Inside a tile the
  // operand reads of k-step s+1 are issued between the MFMAs of k-step s (one wave per SIMD: nobody else
  // hides the LDS latency).
  int it = 0;
  HRP_STAMP(1);
  if (gxi < t.ntiles) issue(gxi, smem);
  // (computed here, under the latency of the first tile's DMA)
  // bf16: the LDS offsets of a lane's operands do not depend on the tile -> registers, NKS k-steps of 16 pixels
  constexpr int NKS_ = NKS > 0 ? NKS : 1;
  int xo0[NKS_], xo1[NKS_], ao[NKS_];
  if constexpr (SZ == 2) {
    auto xoff = [&](int m) {
      int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      if (ti >= t.TI) ti = t.TI - 1;
      return mul24(mul24(mul24(ti, t.IHt) + mul24(ty, IS), t.IWt) + mul24(tx, IS), P) + tr_coff;
    };
#pragma unroll
    for (int ks = 0; ks < NKS_; ++ks) {
      // k-slot (khalf, q4, j) <-> pixel  wave*ppw + 16 ks + 8*khalf + 4*q4 + j
      const int m0 = wave * ppw + ks * 16 + 8 * khalf + tr_pix;
      xo0[ks] = xoff(m0); xo1[ks] = xoff(m0 + 4);
      ao[ks] = m0 * P + tr_coff;
    }
  }

  HRP_STAMP(2);
  for (int tile = gxi; tile < t.ntiles; tile += t.G, ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (it < 4) HRP_STAMP(3 + 2 * it);
    const char* lds_x = smem + (it & 1) * t.buf_bytes;
    const char* lds_dy = lds_x + x_pieces * 1024;
    const bool more = tile + t.G < t.ntiles;
    if constexpr (SZ == 2) {
      // the next tile's DMA pieces are issued between the MFMAs (SPP slots per MFMA) so that a full memory
      // queue stalls the wave while the matrix pipe still has work
      constexpr int SPP = (MAXP_X + MAXP_DY + NKS_ * NTE - 1) / (NKS_ * NTE);
      TileCtx nx{};
      if (more) nx = tile_ctx(tile + t.G, smem + ((it + 1) & 1) * t.buf_bytes);
      if constexpr (NB == 2) {
        // 1x1 layer, 64 x 64 block: per k-step 2 dY fragments x 2 X fragments -> 4 MFMAs; the fragments of
        // k-step ks + 1 are read under the MFMAs of k-step ks
        bf16x8 fa[2][2], fb[2][2];
        auto ld2 = [&](int ks, bf16x8 (&a)[2], bf16x8 (&b)[2]) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            bf16x4 lo = WG<T>::tr(lds_dy + ao[ks] + k * 32 * SZ), hi = WG<T>::tr(lds_dy + ao[ks] + k * 32 * SZ + 4 * P);
            a[k] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            bf16x4 l2 = WG<T>::tr(lds_x + xo0[ks] + tapoff[0] + k * 32 * SZ), h2 = WG<T>::tr(lds_x + xo1[ks] + tapoff[0] + k * 32 * SZ);
            b[k] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        };
        ld2(0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < NKS_; ++ks) {
          const int c = ks & 1, n = c ^ 1;
          if (ks + 1 < NKS_) ld2(ks + 1, fa[n], fb[n]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            WG<T>::mma(fa[c][e >> 1], fb[c][e & 1], acc[e]);
            if (more) {
#pragma unroll
              for (int u = 0; u < SPP; ++u) issue_slot(nx, (ks * 4 + e) * SPP + u);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      // Flattened MFMA sequence q = ks * NT + tp of the tile.  The X fragment of MFMA q + D is read while MFMA q
      // runs (ring of D + 1 fragments: a full next-k-step prefetch would cost 36 more registers and with them
      // the second wave per SIMD that lets this kernel share a CU with the kernels of other lanes); the dY
      // fragment of k-step ks + 1 is read at the start of k-step ks.
      constexpr int D = NT >= 4 ? 4 : NT, TOT = NKS_ * NT;
      bf16x8 a[2], b[D + 1];
      auto load_a = [&](int ks, bf16x8& f) {
        bf16x4 lo = WG<T>::tr(lds_dy + ao[ks]), hi = WG<T>::tr(lds_dy + ao[ks] + 4 * P);
        f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      };
      auto load_b = [&](int q, bf16x8& f) {
        const int ks = q / NT, tp = q % NT;   // constants after unrolling
        bf16x4 lo = WG<T>::tr(lds_x + xo0[ks] + tapoff[tp]), hi = WG<T>::tr(lds_x + xo1[ks] + tapoff[tp]);
        f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      };
      load_a(0, a[0]);
#pragma unroll
      for (int q = 0; q < D && q < TOT; ++q) load_b(q, b[q % (D + 1)]);
#pragma unroll
      for (int q = 0; q < TOT; ++q) {
        const int ks = q / NT, tp = q % NT;
        if (tp == 0 && ks + 1 < NKS_) load_a(ks + 1, a[(ks + 1) & 1]);
        if (q + D < TOT) load_b(q + D, b[(q + D) % (D + 1)]);
        WG<T>::mma(a[ks & 1], b[q % (D + 1)], acc[tp]);
        if (more) {
#pragma unroll
          for (int u = 0; u < SPP; ++u) issue_slot(nx, q * SPP + u);
        }
        // keep the prefetch distance: without the fence the scheduler sinks the reads next to their MFMA
        __builtin_amdgcn_sched_barrier(0);
      }
      }   // NB == 1
    } else {
      if (more) issue(tile + t.G, smem + ((it + 1) & 1) * t.buf_bytes);
      auto load = [&](int kb, float& a, float (&b)[NT]) {
        const int m0 = wave * ppw + kb + khalf;
        a = *(const float*)(lds_dy + m0 * P + l31 * 4);
        const int xo = xtab[m0] + l31 * 4;
#pragma unroll
        for (int tp = 0; tp < NT; ++tp) b[tp] = *(const float*)(lds_x + xo + tapoff[tp]);
      };
      float a_c, b_c[NT];
      load(0, a_c, b_c);
      for (int kb = 0; kb < ppw; kb += K) {
        float a_n, b_n[NT];
        load(kb + K < ppw ? kb + K : kb, a_n, b_n);
#pragma unroll
        for (int tp = 0; tp < NT; ++tp) WG<T>::mma(a_c, b_c[tp], acc[tp]);
        a_c = a_n;
#pragma unroll
        for (int tp = 0; tp < NT; ++tp) b_c[tp] = b_n[tp];
      }
    }
    if (it < 4) HRP_STAMP(4 + 2 * it);
  }

  // ---- combine the 4 waves' partial sums through LDS, half of the accumulator rows at a time: every wave
  // drops its 16 rows x 32 columns per tap in output order, then the 256 threads add the 4 copies as float4
  // and write the partial slab (or, without a workspace, fp32 atomics into dW[co][ci][tap]) ------------------
  __syncthreads();
  HRP_STAMP(11);
  float* dump = (float*)smem;
  const int tstride = d.dw_tap_stride > 0 ? d.dw_tap_stride : d.ntaps;
  // partial slab [g][block][NT*1024], coalesced; a second launch folds the G slabs into dW
  float* ws = t.use_ws ? (float*)d.workspace + ((size_t)gxi * npairs + blk) * (NTE * 1024) : nullptr;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
    float* mine = dump + wave * (NTE * 512) + 4 * khalf * 32 + l31;
#pragma unroll
    for (int tp = 0; tp < NTE; ++tp)
#pragma unroll
      for (int j = 0; j < 8; ++j)   // MFMA register 8h + j -> row 16h + (j & 3) + 8 (j >> 2) + 4 khalf
        mine[(tp * 16 + (j & 3) + 8 * (j >> 2)) * 32] = acc[tp][8 * h + j];
    __syncthreads();
    for (int f = tid; f < NTE * 128; f += 256) {
      float4 v = ((const float4*)dump)[f];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 u = ((const float4*)(dump + w * (NTE * 512)))[f];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      const int tp = f >> 7, rem = f & 127;
      if (ws) {
        ((float4*)(ws + tp * 1024 + 512 * h))[rem] = v;
      } else {
        const int tap = tp / (NB * NB), cbk = (tp % (NB * NB)) / NB, ibk = tp % NB;
        const int co = co0 + 32 * cbk + 16 * h + (rem >> 3), cin = ci0 + 32 * ibk + 4 * (rem & 7);
        if (co < d.Cout) {
          float* o = d.dw + ((size_t)co * d.dw_cin + cin) * tstride + d.dw_tap_off + tap;
          if (cin < d.dw_cin) atomicAdd(o, v.x);
          if (cin + 1 < d.dw_cin) atomicAdd(o + tstride, v.y);
          if (cin + 2 < d.dw_cin) atomicAdd(o + 2 * tstride, v.z);
          if (cin + 3 < d.dw_cin) atomicAdd(o + 3 * tstride, v.w);
        }
      }
    }
    HRP_STAMP(12 + h);
  }
  }   // (not fp32x3)
}

template <typename T, int NT, int NKS, int NB>
__global__ __launch_bounds__(256) WGRAD_OCC void conv_wgrad_kernel(const hrp_wgrad_desc d, const WgradTiling t) {
  conv_wgrad_body<T, NT, NKS, NB>(d, t, blockIdx.x, blockIdx.y);
}

// ---- eight-wave workgroups for the 3x3 stride-1 bf16 layers of the BasicBlocks (32 -> 32 and multiples of 64 channels) -----
// What the batched weight-gradient launches cost is bytes: (tile bytes staged into LDS + 2 x partial-slab bytes) / ~4 TB/s fits
// every class measured with tools/bench_batch.py.  The 32 x 32 program stages the x / dY slices of a layer
// (n_cob + n_cib) / 2 times - 2 / 4 / 8 x for 64 / 128 / 256 channels - and writes one 36 KB slab per four waves.  Here a
// workgroup has EIGHT waves, each running the 32 x 32 wave program (nine taps, 64 pixels per tile, 144 accumulator registers,
// two waves per SIMD as before), arranged as
//   PAIRS = 4:  a 64 x 64 block = 2 x 2 (cout, cin) pairs x 2 pixel slices, 128-pixel tiles: every staged byte feeds twice
//               the MFMAs (the slices are staged (n_cob + n_cib) / 4 times), slab bytes per wave x 2;
//   PAIRS = 1:  one 32 x 32 pair x 8 pixel slices, 512-pixel tiles (the 32-channel layers): slab bytes per wave / 2.
// Stride-2 layers with multiples of 64 channels (fuse-layer down paths, transitions, the cls head's downsamp_modules) run
// PAIRS = 4 on 64-pixel tiles (NKS = 2: the input halo of a 128-pixel tile does not fit): the four-wave program staged a 22 KB
// tile per nine MFMAs of a wave there; step 34.65 -> 34.15 ms.
// LDS: the X halo tile and the dY tile as 32-channel planes laid out exactly like the tiles of conv_wgrad_body (64-byte pixel
// rows: the transpose reads stay conflict free), double buffered.  The partial sums leave in the 32 x 32 slab layout, so the
// folding launches do not change.
constexpr int OCTO_MAXP_X = 5, OCTO_MAXP_DY = 4;

template <int PAIRS, int NKS = 4>
__device__ __forceinline__ void conv_wgrad_octo_body(const hrp_wgrad_desc& d, const WgradTiling& t, const int gxi, const int blk) {
  using T = bf16_t;
  constexpr int NT = 9, SZ = 2, VEC = 8, P = 64, NVEC = 4;
  constexpr int KSL = 8 / PAIRS;               // pixel slices of 16 NKS pixels per tile (NKS = 2: the 64-pixel tiles of stride-2 layers)
  constexpr int BM = 16 * NKS * KSL;
  constexpr int NPL = PAIRS == 4 ? 2 : 1;      // 32-channel planes per operand
  constexpr int DPP = BM / 16;                 // 1 KiB pieces per dY plane
  constexpr int MAXP_X = OCTO_MAXP_X, MAXP_DY = OCTO_MAXP_DY;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0 .. 7 = slice * PAIRS + pair
  const int pair = wave % PAIRS, ksl = wave / PAIRS;
  const int pco = pair >> 1, pci = pair & 1;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int cobw = fdiv(blk, t.fd_cib), cibw = blk - cobw * t.n_cib;
  const int co0 = cobw * (32 * NPL), ci0 = cibw * (32 * NPL);
  const int thw = t.TH * t.TW, ihw = t.IHt * t.IWt;
  const int IS = d.in_stride;
  const int xpp = NPL == 2 ? t.x_pieces >> 1 : t.x_pieces;      // 1 KiB pieces per X plane

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const char* xg = (const char*)d.x;
  const char* dyg = (const char*)d.dy;
  const char* zero = (const char*)g_wg_zero_page;
  asm volatile("" : "+v"(zero));

  int tapoff[NT];
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) tapoff[tp] = ((d.dy_t[tp] - t.mindy) * t.IWt + (d.dx_t[tp] - t.mindx)) * P;

  // DMA plan: as in conv_wgrad_body, per plane; piece p = wave + 8 i of this wave's slot i
  const int x_pieces = t.x_pieces, dy_pieces = t.dy_pieces;
  int xrel[MAXP_X], xcode[MAXP_X], dyrel[MAXP_DY], dycode[MAXP_DY];
  {
    const int y_last = (t.tiles_y - 1) * t.TH, x_last = (t.tiles_x - 1) * t.TW, n_last = (t.tiles_n - 1) * t.TI;
    const int iy_last = y_last * IS + t.mindy, ix_last = x_last * IS + t.mindx;
#pragma unroll
    for (int i = 0; i < MAXP_X; ++i) {
      xcode[i] = 32; xrel[i] = 0;
      const int p = wave + 8 * i;
      if (p >= x_pieces) continue;
      const int plane = (NPL == 2 && p >= xpp) ? 1 : 0, pp = p - plane * xpp;
      const int sl = pp * 64 + lane, pix = sl / NVEC, vec = sl - pix * NVEC;
      const int ti = fdiv16(pix, t.fd_ihw), rem = pix - mul24(ti, ihw);
      const int iy = fdiv16(rem, t.fd_iwt), ix = rem - mul24(iy, t.IWt);
      const int c = ci0 + plane * 32 + vec * VEC;
      int code = (pix >= t.in_pix) ? 32 : 0;
      code |= (iy + t.mindy < 0) ? 1 : 0;
      code |= (iy + iy_last >= d.H) ? 2 : 0;
      code |= (ix + t.mindx < 0) ? 4 : 0;
      code |= (ix + ix_last >= d.W) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      xcode[i] = code;
      xrel[i] = mul24(mul24(mul24(ti, d.H) + iy, d.W) + ix, d.x_pitch * SZ) + c * SZ;
    }
#pragma unroll
    for (int i = 0; i < MAXP_DY; ++i) {
      dycode[i] = 32; dyrel[i] = 0;
      const int p = wave + 8 * i;
      if (p >= dy_pieces) continue;
      const int plane = p / DPP, pp = p - plane * DPP;
      const int sl = pp * 64 + lane, m = sl / NVEC, vec = sl - m * NVEC;
      const int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      const int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      const int c = co0 + plane * 32 + vec * VEC;
      int code = (ti >= t.TI) ? 32 : 0;
      code |= (ty + y_last >= d.Ho) ? 2 : 0;
      code |= (tx + x_last >= d.Wo) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      dycode[i] = code;
      dyrel[i] = mul24(mul24(mul24(ti, d.Ho) + ty, d.Wo) + tx, d.dy_pitch * SZ) + c * SZ;
    }
  }

  struct TileCtx { const char* xbase; const char* dybase; char* buf; int cls; };
  auto tile_ctx = [&](int tile, char* buf) {
    int q = fdiv(tile, t.fd_tx);
    const int tx_i = tile - q * t.tiles_x;
    const int tn_i = fdiv(q, t.fd_ty);
    const int ty_i = q - tn_i * t.tiles_y;
    const int n0 = tn_i * t.TI, oy0 = ty_i * t.TH, ox0 = tx_i * t.TW;
    const int iy0 = oy0 * IS + t.mindy, ix0 = ox0 * IS + t.mindx;
    TileCtx c;
    c.cls = 32 | (ty_i == 0 ? 1 : 0) | (ty_i == t.tiles_y - 1 ? 2 : 0) | (tx_i == 0 ? 4 : 0) |
            (tx_i == t.tiles_x - 1 ? 8 : 0) | (tn_i == t.tiles_n - 1 ? 16 : 0);
    c.xbase = xg + (((long long)n0 * d.H + iy0) * d.W + ix0) * (long long)d.x_pitch * SZ;
    c.dybase = dyg + (((long long)n0 * d.Ho + oy0) * d.Wo + ox0) * (long long)d.dy_pitch * SZ;
    c.buf = buf;
    return c;
  };
  auto issue_slot = [&](const TileCtx& c, int slot) {   // slot is a constant after unrolling
    if (slot < MAXP_X) {
      const int p = wave + 8 * slot;
      if (p < x_pieces) wg_dma16((xcode[slot] & c.cls) ? zero : c.xbase + (unsigned)xrel[slot], c.buf + p * 1024);
    } else if (slot < MAXP_X + MAXP_DY) {
      const int i = slot - MAXP_X, p = wave + 8 * i;
      if (p < dy_pieces)
        wg_dma16((dycode[i] & c.cls) ? zero : c.dybase + (unsigned)dyrel[i], c.buf + (x_pieces + p) * 1024);
    }
  };

  const int tr_pix = (lane & 15) >> 2;
  const int tr_coff = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * SZ;

  int it = 0;
  if (gxi < t.ntiles) {
    const TileCtx c = tile_ctx(gxi, smem);
#pragma unroll
    for (int slot = 0; slot < MAXP_X + MAXP_DY; ++slot) issue_slot(c, slot);
  }
  int xo0[NKS], xo1[NKS], ao[NKS];
  {
    auto xoff = [&](int m) {
      int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      if (ti >= t.TI) ti = t.TI - 1;       // idle slot (its dY row is zero)
      return mul24(mul24(mul24(ti, t.IHt) + mul24(ty, IS), t.IWt) + mul24(tx, IS), P) + tr_coff;
    };
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int m0 = ksl * (16 * NKS) + ks * 16 + 8 * khalf + tr_pix;
      xo0[ks] = xoff(m0); xo1[ks] = xoff(m0 + 4);
      ao[ks] = m0 * P + tr_coff;
    }
  }

  for (int tile = gxi; tile < t.ntiles; tile += t.G, ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* lds_x = smem + (it & 1) * t.buf_bytes + pci * (xpp * 1024);
    const char* lds_dy = smem + (it & 1) * t.buf_bytes + x_pieces * 1024 + pco * (DPP * 1024);
    const bool more = tile + t.G < t.ntiles;
    TileCtx nx{};
    if (more) nx = tile_ctx(tile + t.G, smem + ((it + 1) & 1) * t.buf_bytes);
    // the wave program of conv_wgrad_body (NB = 1): flattened MFMA sequence q = ks * 9 + tap, X fragments D steps ahead, the dY
    // fragment of the next k-step at the first tap; one DMA piece of the next tile every fourth (second) step
    constexpr int D = 4, TOT = NKS * NT, STRIDE = TOT / (MAXP_X + MAXP_DY);
    bf16x8 a[2], b[D + 1];
    auto load_a = [&](int ks, bf16x8& f) {
      bf16x4 lo = WG<T>::tr(lds_dy + ao[ks]), hi = WG<T>::tr(lds_dy + ao[ks] + 4 * P);
      f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_b = [&](int q, bf16x8& f) {
      const int ks = q / NT, tp = q % NT;
      bf16x4 lo = WG<T>::tr(lds_x + xo0[ks] + tapoff[tp]), hi = WG<T>::tr(lds_x + xo1[ks] + tapoff[tp]);
      f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    load_a(0, a[0]);
#pragma unroll
    for (int q = 0; q < D; ++q) load_b(q, b[q % (D + 1)]);
#pragma unroll
    for (int q = 0; q < TOT; ++q) {
      const int ks = q / NT, tp = q % NT;
      if (tp == 0 && ks + 1 < NKS) load_a(ks + 1, a[(ks + 1) & 1]);
      if (q + D < TOT) load_b(q + D, b[(q + D) % (D + 1)]);
      WG<T>::mma(a[ks & 1], b[q % (D + 1)], acc[tp]);
      if (more && q % STRIDE == 0) issue_slot(nx, q / STRIDE);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the partial slabs [g][32 x 32 pair][9 * 1024], half of the accumulator rows at a time: every wave drops its rows in
  // output order, then the 512 threads add the pixel slices of every pair as float4 and write its slab
  __syncthreads();
  float* dump = (float*)smem;
  const int ncib32 = t.n_cib * NPL, np32 = t.n_cob * NPL * ncib32;
  float* ws = (float*)d.workspace + (size_t)gxi * np32 * (NT * 1024);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
    float* mine = dump + wave * (NT * 512) + 4 * khalf * 32 + l31;
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
      for (int j = 0; j < 8; ++j) mine[(tp * 16 + (j & 3) + 8 * (j >> 2)) * 32] = acc[tp][8 * h + j];
    __syncthreads();
    for (int f = tid; f < PAIRS * NT * 128; f += 512) {
      const int pr = f / (NT * 128), e = f - pr * (NT * 128);
      float4 v = ((const float4*)dump)[f];
#pragma unroll
      for (int k = 1; k < KSL; ++k) {
        const float4 u = ((const float4*)dump)[f + k * PAIRS * (NT * 128)];
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      const int cob32 = cobw * NPL + (pr >> 1), cib32 = cibw * NPL + (pr & 1);
      const int tp = e >> 7, rem = e & 127;
      ((float4*)(ws + (size_t)(cob32 * ncib32 + cib32) * (NT * 1024) + tp * 1024 + 512 * h))[rem] = v;
    }
  }
}

// ---- the eight-wave program for HRP_F32X3 (64 x 64 blocks: multiples of 64 channels) ---------------------------------------
// conv_wgrad_octo_body<4> with the staging of conv_wgrad_x3_body: the fp32 tile (64 channels of x and of dY per pixel) lands in a
// RAW region, is converted once into bf16 hi / lo planes of 32 channels, and every wave runs the hi / lo wave program on its
// (cout, cin) pair and pixel slice.  Against the four-wave fp32x3 program a staged byte feeds twice the MFMAs - fp32 tiles are
// twice the bytes, and the weight gradient is paced by its tile fills.  NKS = 4: 128-pixel tiles; 2: 64-pixel tiles (two 8 x 8
// images per tile would need 164 KB).
constexpr int OCTO3_MAXP_X = 7, OCTO3_MAXP_DY = 4;

template <int NT, int NKS>
__device__ __forceinline__ void conv_wgrad_octo_x3_body(const hrp_wgrad_desc& d, const WgradTiling& t, const int gxi, const int blk) {
  constexpr int SZ = 4, VEC = 4, P2 = 64, NVEC = 16;   // RAW pixel row: 64 fp32 = 16 vectors of 16 bytes
  constexpr int BM = 2 * NKS * 16;
  constexpr int MAXP_X = OCTO3_MAXP_X, MAXP_DY = OCTO3_MAXP_DY;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0 .. 7 = slice * 4 + pair
  const int pair = wave & 3, ksl = wave >> 2;
  const int pco = pair >> 1, pci = pair & 1;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int cobw = fdiv(blk, t.fd_cib), cibw = blk - cobw * t.n_cib;
  const int co0 = cobw * 64, ci0 = cibw * 64;
  const int thw = t.TH * t.TW, ihw = t.IHt * t.IWt;
  const int IS = d.in_stride;

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const char* xg = (const char*)d.x;
  const char* dyg = (const char*)d.dy;
  const char* zero = (const char*)g_wg_zero_page;
  asm volatile("" : "+v"(zero));

  int tapoff[NT];
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) tapoff[tp] = ((d.dy_t[tp] - t.mindy) * t.IWt + (d.dx_t[tp] - t.mindx)) * P2;

  const int x_pieces = t.x_pieces, dy_pieces = t.dy_pieces;
  int xrel[MAXP_X], xcode[MAXP_X], dyrel[MAXP_DY], dycode[MAXP_DY];
  {
    const int y_last = (t.tiles_y - 1) * t.TH, x_last = (t.tiles_x - 1) * t.TW, n_last = (t.tiles_n - 1) * t.TI;
    const int iy_last = y_last * IS + t.mindy, ix_last = x_last * IS + t.mindx;
#pragma unroll
    for (int i = 0; i < MAXP_X; ++i) {
      xcode[i] = 32; xrel[i] = 0;
      const int p = wave + 8 * i;
      if (p >= x_pieces) continue;
      const int sl = p * 64 + lane, pix = sl / NVEC, vec = sl - pix * NVEC;
      const int ti = fdiv16(pix, t.fd_ihw), rem = pix - mul24(ti, ihw);
      const int iy = fdiv16(rem, t.fd_iwt), ix = rem - mul24(iy, t.IWt);
      int code = (pix >= t.in_pix) ? 32 : 0;
      code |= (iy + t.mindy < 0) ? 1 : 0;
      code |= (iy + iy_last >= d.H) ? 2 : 0;
      code |= (ix + t.mindx < 0) ? 4 : 0;
      code |= (ix + ix_last >= d.W) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      xcode[i] = code;
      xrel[i] = mul24(mul24(mul24(ti, d.H) + iy, d.W) + ix, d.x_pitch * SZ) + (ci0 + vec * VEC) * SZ;
    }
#pragma unroll
    for (int i = 0; i < MAXP_DY; ++i) {
      dycode[i] = 32; dyrel[i] = 0;
      const int p = wave + 8 * i;
      if (p >= dy_pieces) continue;
      const int sl = p * 64 + lane, m = sl / NVEC, vec = sl - m * NVEC;
      const int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      const int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      int code = (ti >= t.TI) ? 32 : 0;
      code |= (ty + y_last >= d.Ho) ? 2 : 0;
      code |= (tx + x_last >= d.Wo) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      dycode[i] = code;
      dyrel[i] = mul24(mul24(mul24(ti, d.Ho) + ty, d.Wo) + tx, d.dy_pitch * SZ) + (co0 + vec * VEC) * SZ;
    }
  }

  struct TileCtx { const char* xbase; const char* dybase; int cls; };
  auto tile_ctx = [&](int tile) {
    int q = fdiv(tile, t.fd_tx);
    const int tx_i = tile - q * t.tiles_x;
    const int tn_i = fdiv(q, t.fd_ty);
    const int ty_i = q - tn_i * t.tiles_y;
    const int n0 = tn_i * t.TI, oy0 = ty_i * t.TH, ox0 = tx_i * t.TW;
    const int iy0 = oy0 * IS + t.mindy, ix0 = ox0 * IS + t.mindx;
    TileCtx c;
    c.cls = 32 | (ty_i == 0 ? 1 : 0) | (ty_i == t.tiles_y - 1 ? 2 : 0) | (tx_i == 0 ? 4 : 0) |
            (tx_i == t.tiles_x - 1 ? 8 : 0) | (tn_i == t.tiles_n - 1 ? 16 : 0);
    c.xbase = xg + (((long long)n0 * d.H + iy0) * d.W + ix0) * (long long)d.x_pitch * SZ;
    c.dybase = dyg + (((long long)n0 * d.Ho + oy0) * d.Wo + ox0) * (long long)d.dy_pitch * SZ;
    return c;
  };
  char* const raw = smem;                          // [x_pieces + dy_pieces] KiB, the tile as it lies in HBM
  char* const pl = smem + t.buf_bytes;             // X: hi cin 0-31 | hi cin 32-63 | lo | lo;  dY: the same for cout
  const int xps = x_pieces * 256, dps = dy_pieces * 256;       // one plane
  auto issue_slot = [&](const TileCtx& c, int slot) {   // slot is a constant after unrolling
    if (slot < MAXP_X) {
      const int p = wave + 8 * slot;
      if (p < x_pieces) wg_dma16((xcode[slot] & c.cls) ? zero : c.xbase + (unsigned)xrel[slot], raw + p * 1024);
    } else if (slot < MAXP_X + MAXP_DY) {
      const int i = slot - MAXP_X, p = wave + 8 * i;
      if (p < dy_pieces) wg_dma16((dycode[i] & c.cls) ? zero : c.dybase + (unsigned)dyrel[i], raw + (x_pieces + p) * 1024);
    }
  };

  const int tr_pix = (lane & 15) >> 2;
  const int tr_coff = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  if (gxi < t.ntiles) {
    const TileCtx c = tile_ctx(gxi);
#pragma unroll
    for (int slot = 0; slot < MAXP_X + MAXP_DY; ++slot) issue_slot(c, slot);
  }
  int xo0[NKS], xo1[NKS], ao[NKS];
  {
    auto xoff = [&](int m) {
      int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      if (ti >= t.TI) ti = t.TI - 1;
      return mul24(mul24(mul24(ti, t.IHt) + mul24(ty, IS), t.IWt) + mul24(tx, IS), P2) + tr_coff;
    };
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int m0 = ksl * (NKS * 16) + ks * 16 + 8 * khalf + tr_pix;
      xo0[ks] = xoff(m0); xo1[ks] = xoff(m0 + 4);
      ao[ks] = m0 * P2 + tr_coff;
    }
  }
  // conversion: one 32-byte granule (8 floats; 8 per pixel) of RAW -> 16 bytes of a hi plane + 16 bytes of its lo plane
  const int ngran_x = t.in_pix * 8, ngran = ngran_x + BM * 8;

  for (int tile = gxi; tile < t.ntiles; tile += t.G) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // the tile has landed in RAW; every wave is done with the planes of the previous tile
    for (int e = tid; e < ngran; e += 512) {
      const bool isx = e < ngran_x;
      const int r = isx ? e : e - ngran_x;
      const int pix = r >> 3, g = r & 7;
      const char* src = raw + (isx ? 0 : x_pieces * 1024) + r * 32;
      const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 16);
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      uint4 hi, lo;
      split_bf16x8(x, hi, lo);
      const int ps = isx ? xps : dps;
      char* dst = pl + (isx ? 0 : 4 * xps) + (g >> 2) * ps + pix * P2 + (g & 3) * 16;
      *(uint4*)dst = hi;
      *(uint4*)(dst + 2 * ps) = lo;
    }
    __syncthreads();
    const bool more = tile + t.G < t.ntiles;
    if (more) {
      const TileCtx nx = tile_ctx(tile + t.G);
#pragma unroll
      for (int slot = 0; slot < MAXP_X + MAXP_DY; ++slot) issue_slot(nx, slot);
    }
    const char* lds_x = pl + pci * xps;
    const char* lds_dy = pl + 4 * xps + pco * dps;
    constexpr int TOT = NKS * NT, D = TOT >= 2 ? 2 : 1;
    bf16x8 ah[2], al[2], bh[D + 1], bl[D + 1];
    auto tr2 = [&](const char* p0, const char* p1) {
      bf16x4 lo = WG<bf16_t>::tr(p0), hi = WG<bf16_t>::tr(p1);
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_a = [&](int ks, bf16x8& fh, bf16x8& fl) {
      fh = tr2(lds_dy + ao[ks], lds_dy + ao[ks] + 4 * P2);
      fl = tr2(lds_dy + 2 * dps + ao[ks], lds_dy + 2 * dps + ao[ks] + 4 * P2);
    };
    auto load_b = [&](int q, bf16x8& fh, bf16x8& fl) {
      const int ks = q / NT, tp = q % NT;   // constants after unrolling
      fh = tr2(lds_x + xo0[ks] + tapoff[tp], lds_x + xo1[ks] + tapoff[tp]);
      fl = tr2(lds_x + 2 * xps + xo0[ks] + tapoff[tp], lds_x + 2 * xps + xo1[ks] + tapoff[tp]);
    };
    load_a(0, ah[0], al[0]);
#pragma unroll
    for (int q = 0; q < D; ++q) load_b(q, bh[q % (D + 1)], bl[q % (D + 1)]);
#pragma unroll
    for (int q = 0; q < TOT; ++q) {
      const int ks = q / NT, tp = q % NT, r = q % (D + 1);
      if (tp == 0 && ks + 1 < NKS) load_a(ks + 1, ah[(ks + 1) & 1], al[(ks + 1) & 1]);
      if (q + D < TOT) load_b(q + D, bh[(q + D) % (D + 1)], bl[(q + D) % (D + 1)]);
      acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks & 1], bh[r], acc[tp], 0, 0, 0);      // (small terms first)
      acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks & 1], bl[r], acc[tp], 0, 0, 0);
      acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks & 1], bh[r], acc[tp], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- the partial slabs, as conv_wgrad_octo_body<4> writes them
  __syncthreads();
  float* dump = (float*)smem;
  const int ncib32 = t.n_cib * 2, np32 = t.n_cob * 2 * ncib32;
  float* ws = (float*)d.workspace + (size_t)gxi * np32 * (NT * 1024);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
    float* mine = dump + wave * (NT * 512) + 4 * khalf * 32 + l31;
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
      for (int j = 0; j < 8; ++j) mine[(tp * 16 + (j & 3) + 8 * (j >> 2)) * 32] = acc[tp][8 * h + j];
    __syncthreads();
    for (int f = tid; f < 4 * NT * 128; f += 512) {
      const int pr = f / (NT * 128), e = f - pr * (NT * 128);
      float4 v = ((const float4*)dump)[f];
      const float4 u = ((const float4*)dump)[f + 4 * (NT * 128)];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      const int cob32 = cobw * 2 + (pr >> 1), cib32 = cibw * 2 + (pr & 1);
      const int tp = e >> 7, rem = e & 127;
      ((float4*)(ws + (size_t)(cob32 * ncib32 + cib32) * (NT * 1024) + tp * 1024 + 512 * h))[rem] = v;
    }
  }
}

// dW[co][ci][tp] (+)= sum_g ws[g][blk][tp][row][ci].  bx: which 256 consecutive elements of the pair's NTE * 1024.
constexpr int FOLD_ELEMS = 256;   // per block: 64 lanes x float4, the 4 waves take every fourth slab
__device__ __forceinline__ void wgrad_fold_body(const hrp_wgrad_fold_desc& f, const int bx, const int blk) {
  const int G = f.G, NTE = f.nte, NB = f.nb;
  const int cob = blk / f.n_cib, cib = blk - cob * f.n_cib;
  const int tstride = f.dw_tap_stride > 0 ? f.dw_tap_stride : f.ntaps;
  __shared__ float4 part[4][64];
  const float* ws = f.workspace + (size_t)blk * (NTE * 1024);
  const size_t gstride = (size_t)f.pairs * (NTE * 1024);
  const int e = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int i = bx * FOLD_ELEMS + 4 * e;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int g = ph;
  for (; g + 28 < G; g += 32) {   // 8 slabs in flight per thread
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(ws + (size_t)(g + 4 * u) * gstride + i);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; g < G; g += 4) {
    const float4 v = *(const float4*)(ws + (size_t)g * gstride + i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  part[ph][e] = s;
  __syncthreads();
  // thread (ph, e) finishes element 4 e + ph: the four phases' partial sums in slab order
  {
    const float* p0 = (const float*)&part[0][e];
    const float* p1 = (const float*)&part[1][e];
    const float* p2 = (const float*)&part[2][e];
    const float* p3 = (const float*)&part[3][e];
    const float r = p0[ph] + p1[ph] + p2[ph] + p3[ph];
    const int ii = i + ph;
    const int ci = ii & 31, row = (ii >> 5) & 31, tp = ii >> 10;
    const int tap = tp / (NB * NB), cbk = (tp % (NB * NB)) / NB, ibk = tp % NB;
    const int co = (cob * NB + cbk) * 32 + row, cin = (cib * NB + ibk) * 32 + ci;
    if (co < f.Cout && cin < f.dw_cin) {
      float* o = &f.dw[((size_t)co * f.dw_cin + cin) * tstride + f.dw_tap_off + tap];
      *o = f.accumulate ? *o + r : r;
    }
  }
}

static inline hrp_wgrad_fold_desc make_fold_desc(const hrp_wgrad_desc& d, int G, int pairs, int n_cib, int NTE, int NB) {
  hrp_wgrad_fold_desc f{};
  f.workspace = (const float*)d.workspace; f.dw = d.dw;
  f.G = G; f.pairs = pairs; f.n_cib = n_cib; f.nte = NTE; f.nb = NB;
  f.Cout = d.Cout; f.dw_cin = d.dw_cin; f.ntaps = d.ntaps; f.dw_tap_stride = d.dw_tap_stride; f.dw_tap_off = d.dw_tap_off;
  f.accumulate = d.accumulate;
  return f;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const hrp_wgrad_fold_desc f) {
  wgrad_fold_body(f, blockIdx.x, blockIdx.y);
}

// wg_total: workgroups this problem may use (0: the single-launch default, about one per CU)
template <typename T, int NT, int NB>
static int wgrad_tiling(const hrp_wgrad_desc& d, WgradTiling& t, int wg_budget = 0) {
  constexpr int SZ = Elem<T>::SZ;
  constexpr int P = 32 * NB * SZ;
  constexpr int NTE = NT * NB * NB;
  int mindy = 1 << 30, maxdy = -(1 << 30), mindx = 1 << 30, maxdx = -(1 << 30);
  for (int i = 0; i < d.ntaps; ++i) {
    mindy = d.dy_t[i] < mindy ? d.dy_t[i] : mindy; maxdy = d.dy_t[i] > maxdy ? d.dy_t[i] : maxdy;
    mindx = d.dx_t[i] < mindx ? d.dx_t[i] : mindx; maxdx = d.dx_t[i] > maxdx ? d.dx_t[i] : maxdx;
  }
  t.mindy = mindy; t.mindx = mindx;
  static const int budget_kb = 72;
  const int budget = budget_kb * 1024;
  int lds = 0;
  constexpr int BM_MIN = std::is_same<T, float>::value ? 16 : 64;   // bf16 / fp32x3 need 16 pixels per wave and k-step; fp32 tiles are twice the bytes
  constexpr int BM_MAX = std::is_same<T, f32x3_t>::value ? 128 : 256;   // (fp32x3: RAW tile + planes = twice the tile in LDS, k-steps 1 / 2)
  for (int BM = BM_MAX; BM >= BM_MIN; BM >>= 1) {
    int TW = 1; while (TW < d.Wo && TW < 16) TW <<= 1;
    int TH = 1; while (TH < d.Ho && TH * TW < BM) TH <<= 1;
    int TI = BM / (TW * TH);
    if (TI > d.N) TI = d.N;
    t.TW = TW; t.TH = TH; t.BM = BM;
    t.IHt = (TH - 1) * d.in_stride + (maxdy - mindy) + 1;
    t.IWt = (TW - 1) * d.in_stride + (maxdx - mindx) + 1;
    {
      int maxti = ((budget - BM * 4) / 2 - BM * P - 2048) / (t.IHt * t.IWt * P);
      if (maxti < 1) maxti = 1;
      if (TI > maxti) TI = maxti;
    }
    t.TI = TI;
    t.in_pix = TI * t.IHt * t.IWt;
    t.x_pieces = cdiv(t.in_pix * P, 1024);
    t.dy_pieces = cdiv(BM * P, 1024);
    t.buf_bytes = (t.x_pieces + t.dy_pieces) * 1024;
    t.lds_tab_off = 2 * t.buf_bytes;
    int main_bytes = t.lds_tab_off + BM * 4;
    if (t.x_pieces > 4 * wgrad_maxp_x<T>() || t.dy_pieces > 24) { lds = 1 << 30; continue; }
    t.lds_red_off = 0;  // the cross-wave exchange reuses the tiles
    int red_bytes = NTE * 8192;
    lds = main_bytes > red_bytes ? main_bytes : red_bytes;
    long pixels = (long)d.N * d.Ho * d.Wo;
    if (lds <= budget + 8192 && (BM == BM_MIN || pixels >= BM)) break;
  }
  if (lds > 160 * 1024) {
    set_error("wgrad: tile does not fit LDS / the DMA plan (H=%d W=%d stride=%d taps=%d)", d.H, d.W, d.in_stride, d.ntaps);
    return HRP_ERR_ARG;
  }
  t.lds_bytes = lds;
  t.tiles_x = cdiv(d.Wo, t.TW); t.tiles_y = cdiv(d.Ho, t.TH); t.tiles_n = cdiv(d.N, t.TI);
  t.ntiles = t.tiles_x * t.tiles_y * t.tiles_n;
  // the DMA plan classifies a lane's validity per tile row / column class: only the first and the last tile
  // row (column) may reach outside the image
  if ((t.tiles_y >= 2 && (t.TH * d.in_stride + mindy < 0 ||
                          (t.tiles_y - 2) * t.TH * d.in_stride + mindy + t.IHt - 1 >= d.H)) ||
      (t.tiles_x >= 2 && (t.TW * d.in_stride + mindx < 0 ||
                          (t.tiles_x - 2) * t.TW * d.in_stride + mindx + t.IWt - 1 >= d.W))) {
    set_error("wgrad: tap offsets reach beyond the border tiles (H=%d W=%d Ho=%d Wo=%d stride=%d)", d.H, d.W, d.Ho, d.Wo,
              d.in_stride);
    return HRP_ERR_ARG;
  }
  t.n_cob = cdiv(d.Cout, 32 * NB); t.n_cib = cdiv(d.Cin, 32 * NB);
  int pairs = t.n_cob * t.n_cib;
  // workgroups per (cout, cin) block: ~2 per CU over the whole launch; each walks ntiles / G pixel tiles
  static const int wg_total = 256;   // (swept: DESIGN 5)
  int G = (wg_budget > 0 ? wg_budget : wg_total) / pairs;
  if (G < 1) G = 1;
  if (G > t.ntiles) G = t.ntiles;
  t.xmode = 0; t.xa = t.xb = t.xnb = 1;
  static const bool no_xcd = false;
  // tuning knob: bit 0 mode 1, bit 1 mode 2.  Measured (PMC FETCH_SIZE of the weight-gradient launches of one step, B = 64):
  // pair-major 27.1 GB, mode 2 only 22.1 GB, mode 1 only 28.5 GB, both 23.5 GB; kernel time one by one 8.31 / 8.75 / 8.38 /
  // 8.78 ms; step time the same within 0.1 ms -> mode 2 (the 2 x 2 blocks of the 64-channel layers share an L2)
  static const int xmask = 2;
  if (wg_budget > 0 && !no_xcd) {          // batched launches (1-D grid): XCD-aware numbering
    if (pairs >= 8 && pairs % 8 == 0 && (xmask & 1)) {
      // xa | n_cob, xb | n_cib, (n_cob / xa) * (n_cib / xb) == 8, xa + xb minimal (slices fetched per XCD)
      int best = 1 << 30;
      for (int a = 1; a <= t.n_cob; ++a) {
        if (t.n_cob % a) continue;
        const int na = t.n_cob / a;
        if (8 % na) continue;
        const int nb_ = 8 / na;
        if (t.n_cib % nb_) continue;
        const int b = t.n_cib / nb_;
        if (a + b < best) { best = a + b; t.xa = a; t.xb = b; t.xnb = nb_; t.xmode = 1; }
      }
    } else if ((pairs == 1 || pairs == 2 || pairs == 4) && G >= 8 && (xmask & 2)) {
      G -= G % 8;
      t.xmode = 2;
    }
  }
  t.G = G;
  t.fd_g = make_fastdiv(G);
  t.fd_ihw = make_fastdiv(t.IHt * t.IWt); t.fd_iwt = make_fastdiv(t.IWt);
  t.fd_thw = make_fastdiv(t.TH * t.TW); t.fd_tw = make_fastdiv(t.TW);
  t.fd_tx = make_fastdiv(t.tiles_x); t.fd_ty = make_fastdiv(t.tiles_y); t.fd_cib = make_fastdiv(t.n_cib);
  return HRP_OK;
}

// Eight-wave program (conv_wgrad_octo_body): 0 = not eligible, else PAIRS.  Depends on the layer only, never on the workspace.
static int octo_pairs(const hrp_wgrad_desc& d) {
  if (d.dtype == HRP_F32 || (d.in_stride != 1 && !(d.in_stride == 2 && d.ntaps == 9))) return 0;
  if (d.ntaps != 9 && !(d.ntaps == 1 && d.dtype == HRP_F32X3 && d.dy_t[0] == 0 && d.dx_t[0] == 0)) return 0;   // (bf16 1x1: NB = 2)
  if (d.dw_cin != d.Cin || d.Ho * d.in_stride != d.H || d.Wo * d.in_stride != d.W || d.dw_tap_stride != 0) return 0;
  for (int i = 0; i < d.ntaps; ++i)
    if (d.dy_t[i] < -1 || d.dy_t[i] > 1 || d.dx_t[i] < -1 || d.dx_t[i] > 1) return 0;
  if (d.Cout == 32 && d.Cin == 32) return d.dtype == HRP_BF16 && d.in_stride == 1 ? 1 : 0;      // (fp32x3, stride 2: the four-wave program)
  return (d.Cout % 64 == 0 && d.Cin % 64 == 0) ? 4 : 0;
}

static int wgrad_tiling_octo(const hrp_wgrad_desc& d, WgradTiling& t, int pairs_w, int wg_budget, int nks = 4) {
  const bool x3 = d.dtype == HRP_F32X3;
  const int npl = pairs_w == 4 ? 2 : 1, BM = nks * 16 * (8 / pairs_w);
  const int halo = d.ntaps == 1 ? 0 : 1;
  t.mindy = -halo; t.mindx = -halo;
  int TW = 1; while (TW < d.Wo && TW < 16) TW <<= 1;
  int TH = 1; while (TH < d.Ho && TH * TW < BM) TH <<= 1;
  int TI = BM / (TW * TH);
  if (TI > d.N) TI = d.N;
  const int IS = d.in_stride;
  t.TW = TW; t.TH = TH; t.TI = TI; t.BM = BM;
  t.IHt = (TH - 1) * IS + 1 + 2 * halo; t.IWt = (TW - 1) * IS + 1 + 2 * halo;
  t.in_pix = TI * t.IHt * t.IWt;
  if (x3) {        // RAW fp32 tile: 64 channels = 256 bytes per pixel; the planes take the same bytes again
    if (pairs_w != 4) return HRP_ERR_ARG;
    t.x_pieces = cdiv(t.in_pix * 256, 1024); t.dy_pieces = BM / 4;
    if (t.x_pieces > 8 * OCTO3_MAXP_X || t.dy_pieces > 8 * OCTO3_MAXP_DY) return HRP_ERR_ARG;
  } else {
    const int xpp = cdiv(t.in_pix * 64, 1024);
    if (npl * xpp > 8 * OCTO_MAXP_X) return HRP_ERR_ARG;          // (tiny maps with many images per tile: the 32 x 32 program)
    t.x_pieces = npl * xpp; t.dy_pieces = npl * (BM / 16);
  }
  t.buf_bytes = (t.x_pieces + t.dy_pieces) * 1024;
  t.lds_tab_off = 0; t.lds_red_off = 0;
  const int red_bytes = 8 * d.ntaps * 512 * 4;
  t.lds_bytes = 2 * t.buf_bytes > red_bytes ? 2 * t.buf_bytes : red_bytes;
  if (t.lds_bytes > 160 * 1024) return HRP_ERR_ARG;
  t.tiles_x = cdiv(d.Wo, TW); t.tiles_y = cdiv(d.Ho, TH); t.tiles_n = cdiv(d.N, TI);
  t.ntiles = t.tiles_x * t.tiles_y * t.tiles_n;
  if ((t.tiles_y >= 2 && (TH * IS - halo < 0 || (t.tiles_y - 2) * TH * IS - halo + t.IHt - 1 >= d.H)) ||
      (t.tiles_x >= 2 && (TW * IS - halo < 0 || (t.tiles_x - 2) * TW * IS - halo + t.IWt - 1 >= d.W)))
    return HRP_ERR_ARG;
  t.n_cob = d.Cout / (32 * npl); t.n_cib = d.Cin / (32 * npl);
  const int pairs = t.n_cob * t.n_cib;
  int G = wg_budget / pairs;
  if (G < 1) G = 1;
  if (G > t.ntiles) G = t.ntiles;
  t.xmode = 0; t.xa = t.xb = t.xnb = 1;
  if ((pairs == 1 || pairs == 2 || pairs == 4) && G >= 8) { G -= G % 8; t.xmode = 2; }
  t.G = G;
  t.use_ws = 1;
  t.fd_g = make_fastdiv(G);
  t.fd_ihw = make_fastdiv(t.IHt * t.IWt); t.fd_iwt = make_fastdiv(t.IWt);
  t.fd_thw = make_fastdiv(TH * TW); t.fd_tw = make_fastdiv(TW);
  t.fd_tx = make_fastdiv(t.tiles_x); t.fd_ty = make_fastdiv(t.tiles_y); t.fd_cib = make_fastdiv(t.n_cib);
  return HRP_OK;
}

// 64 x 64 blocks for the 1x1 bf16 layers with at least 64 channels on both sides
template <typename T, int NT>
static constexpr bool can_nb2() { return NT == 1 && Elem<T>::SZ == 2; }
static inline bool want_nb2(const hrp_wgrad_desc& d) { return d.Cout >= 64 && d.Cin >= 64 && d.dw_cin >= 64; }

template <typename T, int NT, int NB>
static int64_t wgrad_ws_bytes_nb(const hrp_wgrad_desc& d) {
  WgradTiling t{};
  if (wgrad_tiling<T, NT, NB>(d, t) != HRP_OK) return 0;
  return (int64_t)t.G * t.n_cob * t.n_cib * (NT * NB * NB) * 1024 * 4;
}
template <typename T, int NT>
static int64_t wgrad_ws_bytes(const hrp_wgrad_desc& d) {
  if constexpr (can_nb2<T, NT>()) {
    if (want_nb2(d)) return wgrad_ws_bytes_nb<T, NT, 2>(d);
  }
  return wgrad_ws_bytes_nb<T, NT, 1>(d);
}

template <typename T, int NT, int NB>
static int launch_wgrad_nb(const hrp_wgrad_desc& d, hipStream_t s) {
  constexpr int NTE = NT * NB * NB;
  WgradTiling t{};
  int rc = wgrad_tiling<T, NT, NB>(d, t);
  if (rc != HRP_OK) return rc;
  const int pairs = t.n_cob * t.n_cib;
  const int64_t need = (int64_t)t.G * pairs * NTE * 1024 * 4;
  t.use_ws = (d.workspace && d.workspace_bytes >= need) ? 1 : 0;
  if (!t.use_ws && !d.accumulate) {
    if (d.dw_tap_stride > 0 && d.dw_tap_stride != d.ntaps) {
      set_error("wgrad: a tap group without a workspace must accumulate (the caller zeroes dw)");
      return HRP_ERR_ARG;
    }
    zero_async(d.dw, sizeof(float) * (size_t)d.Cout * d.dw_cin * d.ntaps, s);
  }
  void (*kern)(const hrp_wgrad_desc, const WgradTiling) = nullptr;
  if constexpr (Elem<T>::SZ == 2 || std::is_same<T, f32x3_t>::value) {
    // bf16 / fp32x3: k-steps of 16 pixels per wave and tile = BM / 64, unrolled at compile time
    if constexpr (std::is_same<T, f32x3_t>::value) kern = t.BM == 128 ? conv_wgrad_kernel<T, NT, 2, NB> : conv_wgrad_kernel<T, NT, 1, NB>;
    else kern = t.BM == 256 ? conv_wgrad_kernel<T, NT, 4, NB> : t.BM == 128 ? conv_wgrad_kernel<T, NT, 2, NB> : conv_wgrad_kernel<T, NT, 1, NB>;
  } else {
    kern = conv_wgrad_kernel<T, NT, 0, NB>;
  }
  static bool attr_set[5] = {};   // per (T, NT, NB) instantiation, indexed by BM / 64
  if (!attr_set[t.BM / 64]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[t.BM / 64] = true;
  }
  hipLaunchKernelGGL(kern, dim3(t.G, pairs), dim3(256), t.lds_bytes, s, d, t);
  rc = check_launch("conv_wgrad_kernel");
  if (rc != HRP_OK || !t.use_ws || d.phase == 1) return rc;   // phase 1: the caller folds the slabs later
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(NTE * 1024 / FOLD_ELEMS, pairs), dim3(256), 0, s, make_fold_desc(d, t.G, pairs, t.n_cib, NTE, NB));
  return check_launch("wgrad_reduce_kernel");
}
template <typename T, int NT>
static int launch_wgrad(const hrp_wgrad_desc& d, hipStream_t s) {
  if constexpr (can_nb2<T, NT>()) {
    if (want_nb2(d)) return launch_wgrad_nb<T, NT, 2>(d, s);
  }
  return launch_wgrad_nb<T, NT, 1>(d, s);
}


// ---- batched launches (hrp_batch_*, include/hrp.h) --------------------------------------------------------------
// n weight-gradient problems of one tap count and element type in one launch, each with its own tiling; a second
// launch folds every problem's partial slabs.  The workgroups of the launch (about two per CU) are shared between the
// problems in proportion to their work, so a problem contributes G = share / pairs slabs instead of the 256 / pairs of
// a launch of its own: the eight 3x3 layers of a stage-4 step write 8 x 2.3 MB of slabs instead of 8 x 9.4 MB.
struct WgradProblem {
  hrp_wgrad_desc d;
  WgradTiling t;
  int nks, nb, nte, pairs;       // nb: 1 / 2 = 32 / 64-channel blocks of conv_wgrad_body; 3 / 6 = the eight-wave program, PAIRS = nb - 2
  FastDiv fd_r;   // division by nte * 4 (blocks per pair of the folding launch)
  int fold_pairs, fold_n_cib, fold_nb;   // the slab layout the folding launches see (eight-wave program: 32 x 32 pairs)
  int wblk0;      // eight-wave program: first workgroup of this problem in its launch
};

template <typename T, int NT>
__global__ __launch_bounds__(256) WGRAD_OCC void wgrad_batch_kernel(const WgradProblem* __restrict__ tab, const BatchHdr h) {
  int base;
  const int g = batch_find(h, blockIdx.x, base);
  const WgradProblem& P = tab[g];
  const int local = (int)blockIdx.x - base;
  int gxi, blk;
  wgrad_block_of(P.t, local, gxi, blk);
  if constexpr (std::is_same<T, float>::value) {
    conv_wgrad_body<T, NT, 0, 1>(P.d, P.t, gxi, blk);
  } else if constexpr (NT == 1 && Elem<T>::SZ == 2) {
    switch (P.nks * 4 + P.nb) {
      case 4 * 4 + 1: conv_wgrad_body<T, NT, 4, 1>(P.d, P.t, gxi, blk); break;
      case 2 * 4 + 1: conv_wgrad_body<T, NT, 2, 1>(P.d, P.t, gxi, blk); break;
      case 1 * 4 + 1: conv_wgrad_body<T, NT, 1, 1>(P.d, P.t, gxi, blk); break;
      case 4 * 4 + 2: conv_wgrad_body<T, NT, 4, 2>(P.d, P.t, gxi, blk); break;
      case 2 * 4 + 2: conv_wgrad_body<T, NT, 2, 2>(P.d, P.t, gxi, blk); break;
      default: conv_wgrad_body<T, NT, 1, 2>(P.d, P.t, gxi, blk); break;
    }
  } else {
    switch (P.nks) {
      case 4: if constexpr (Elem<T>::SZ == 2) { conv_wgrad_body<T, NT, 4, 1>(P.d, P.t, gxi, blk); } break;
      case 2: conv_wgrad_body<T, NT, 2, 1>(P.d, P.t, gxi, blk); break;
      default: conv_wgrad_body<T, NT, 1, 1>(P.d, P.t, gxi, blk); break;
    }
  }
}

// the eight-wave problems of a batch (nb >= 3): 512 threads per workgroup, so a launch of their own next to the others' launch
__global__ __launch_bounds__(512) void wgrad_octo_batch_kernel(const WgradProblem* __restrict__ tab, const int n) {
  int g = 0, base = 0;
  for (int i = 0; i < n; ++i) {
    const int b0 = tab[i].wblk0;
    if (tab[i].nb >= 3 && (int)blockIdx.x >= b0) { g = i; base = b0; }
  }
  const WgradProblem& P = tab[g];
  int gxi, blk;
  wgrad_block_of(P.t, (int)blockIdx.x - base, gxi, blk);
  if (P.nb == 6 && P.nks == 2) conv_wgrad_octo_body<4, 2>(P.d, P.t, gxi, blk);
  else if (P.nb == 6) conv_wgrad_octo_body<4>(P.d, P.t, gxi, blk);
  else conv_wgrad_octo_body<1>(P.d, P.t, gxi, blk);
}

template <int NT>
__global__ __launch_bounds__(512) void wgrad_octo_x3_batch_kernel(const WgradProblem* __restrict__ tab, const int n) {
  int g = 0, base = 0;
  for (int i = 0; i < n; ++i) {
    const int b0 = tab[i].wblk0;
    if (tab[i].nb >= 3 && (int)blockIdx.x >= b0) { g = i; base = b0; }
  }
  const WgradProblem& P = tab[g];
  int gxi, blk;
  wgrad_block_of(P.t, (int)blockIdx.x - base, gxi, blk);
  if (P.nks == 4) conv_wgrad_octo_x3_body<NT, 4>(P.d, P.t, gxi, blk);
  else if (P.nks == 2) conv_wgrad_octo_x3_body<NT, 2>(P.d, P.t, gxi, blk);
  else if constexpr (NT == 9) conv_wgrad_octo_x3_body<NT, 1>(P.d, P.t, gxi, blk);     // (stride-2 layers: 32-pixel tiles)
}

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradProblem* __restrict__ tab, const BatchHdr h) {
  int base;
  const int g = batch_find(h, blockIdx.x, base);
  const WgradProblem& P = tab[g];
  const int local = (int)blockIdx.x - base;
  const int blk = fdiv(local, P.fd_r);
  const int bx = local - blk * (P.nte * 4);
  hrp_wgrad_fold_desc f;
  f.workspace = (const float*)P.d.workspace; f.dw = P.d.dw;
  f.G = P.t.G; f.pairs = P.fold_pairs; f.n_cib = P.fold_n_cib; f.nte = P.nte; f.nb = P.fold_nb;
  f.Cout = P.d.Cout; f.dw_cin = P.d.dw_cin; f.ntaps = P.d.ntaps; f.dw_tap_stride = P.d.dw_tap_stride; f.dw_tap_off = P.d.dw_tap_off;
  f.accumulate = P.d.accumulate;
  wgrad_fold_body(f, bx, blk);
}

// ---- the deferred fold of up to HRP_BATCH_MAX phase-1 problems in one launch (any tap counts / element types) ----
struct FoldProblem {
  hrp_wgrad_fold_desc f;
  FastDiv fd_r;   // division by nte * 4
  int pad[2];
};

__global__ __launch_bounds__(256) void wgrad_fold_batch_kernel(const FoldProblem* __restrict__ tab, const BatchHdr h) {
  int base;
  const int g = batch_find(h, blockIdx.x, base);
  const FoldProblem& P = tab[g];
  const int local = (int)blockIdx.x - base;
  const int blk = fdiv(local, P.fd_r);
  const int bx = local - blk * (P.f.nte * 4);
  wgrad_fold_body(P.f, bx, blk);
}

static int wgrad_check(const hrp_wgrad_desc* d) {
  HRP_REQUIRE(d && d->x && d->dy && d->dw, "wgrad: null pointer");
  HRP_REQUIRE(d->dtype == HRP_F32 || d->dtype == HRP_BF16 || d->dtype == HRP_F32X3, "wgrad: dtype");
  const int vec = d->dtype == HRP_BF16 ? 8 : 4;
  HRP_REQUIRE(d->Cin % vec == 0 && d->x_pitch % vec == 0 && (uintptr_t)d->x % 16 == 0, "wgrad: x channels/pitch/alignment");
  // dy may have any Cout as long as whole 16-byte vectors can be read (garbage lanes are masked at the store)
  HRP_REQUIRE(d->dy_pitch % vec == 0 && d->dy_pitch >= (d->Cout + vec - 1) / vec * vec && (uintptr_t)d->dy % 16 == 0,
              "wgrad: dy pitch must be a multiple of %d and cover Cout rounded up (Cout=%d pitch=%d)", vec, d->Cout, d->dy_pitch);
  HRP_REQUIRE(d->ntaps == 1 || d->ntaps == 4 || d->ntaps == 9, "wgrad: ntaps=%d unsupported", d->ntaps);
  HRP_REQUIRE(d->dw_cin <= d->Cin, "wgrad: dw_cin > Cin");
  HRP_REQUIRE(d->dw_tap_stride == 0 || (d->dw_tap_off >= 0 && d->dw_tap_off + d->ntaps <= d->dw_tap_stride), "wgrad: tap group");
  return HRP_OK;
}

template <typename T, int NT>
static int wgrad_plan_one(const hrp_wgrad_desc& d, WgradProblem& P, int budget, int octo = 0) {
  if (octo) {
    int rc = wgrad_tiling_octo(d, P.t, octo, budget > 0 ? budget : 256);
    P.nks = 4;
    if (rc != HRP_OK && (d.dtype == HRP_F32X3 || (d.in_stride == 2 && octo == 4))) { rc = wgrad_tiling_octo(d, P.t, octo, budget > 0 ? budget : 256, 2); P.nks = 2; }
    if (rc != HRP_OK && d.dtype == HRP_F32X3 && d.in_stride == 2) { rc = wgrad_tiling_octo(d, P.t, octo, budget > 0 ? budget : 256, 1); P.nks = 1; }
    if (rc != HRP_OK) return rc;
    P.nb = 2 + octo; P.nte = NT;
    P.pairs = P.t.n_cob * P.t.n_cib;
    P.fold_pairs = (d.Cout / 32) * (d.Cin / 32); P.fold_n_cib = d.Cin / 32; P.fold_nb = 1;
    P.fd_r = make_fastdiv(P.nte * 4);
    return HRP_OK;
  }
  P.nb = 1;
  int rc;
  if constexpr (can_nb2<T, NT>()) {
    if (want_nb2(d)) P.nb = 2;
  }
  if (P.nb == 2) {
    if constexpr (can_nb2<T, NT>()) rc = wgrad_tiling<T, NT, 2>(d, P.t, budget);
    else rc = HRP_ERR_ARG;
  } else {
    rc = wgrad_tiling<T, NT, 1>(d, P.t, budget);
  }
  if (rc != HRP_OK) return rc;
  P.nks = std::is_same<T, float>::value ? 0 : P.t.BM / 64;
  P.nte = NT * P.nb * P.nb;
  P.pairs = P.t.n_cob * P.t.n_cib;
  P.fold_pairs = P.pairs; P.fold_n_cib = P.t.n_cib; P.fold_nb = P.nb;
  P.fd_r = make_fastdiv(P.nte * 4);
  return HRP_OK;
}

template <typename T, int NT>
static int wgrad_batch_prepare_nt(const hrp_wgrad_desc* descs, int n, WgradProblem* tab, hrp_batch_info* info) {
  WgradProblem probs[HRP_BATCH_MAX];
  double work[HRP_BATCH_MAX], total[2] = {0.0, 0.0};
  int octo[HRP_BATCH_MAX];
  for (int i = 0; i < n; ++i) {
    memset(&probs[i], 0, sizeof(WgradProblem));
    probs[i].d = descs[i];
    octo[i] = 0;
    if constexpr (!std::is_same<T, float>::value && (NT == 9 || (NT == 1 && std::is_same<T, f32x3_t>::value))) {
      octo[i] = octo_pairs(descs[i]);
      if (octo[i] && wgrad_plan_one<T, NT>(descs[i], probs[i], 0, octo[i]) != HRP_OK) octo[i] = 0;
    }
    if (!octo[i]) {
      const int rc = wgrad_plan_one<T, NT>(descs[i], probs[i], 0);
      if (rc != HRP_OK) return rc;
    }
    work[i] = (double)descs[i].N * descs[i].Ho * descs[i].Wo * probs[i].fold_pairs;
    total[octo[i] ? 1 : 0] += work[i];
  }
  // workgroups of the launch: two per CU, shared in proportion to the work (at least one per (cout, cin) pair); the eight-wave
  // problems' launch: one per CU
  static const int wg_launch = 512;
  static const int wg_octo = 256;   // (512: a batch of eight 85.9 -> 74.2 us instead of 63.6 - the slab bytes double)
  int lds_max = 0, lds_octo = 0, blk = 0, blk2 = 0, blkw = 0;
  for (int i = 0; i < n; ++i) {
    WgradProblem& P = probs[i];
    int budget = (int)((octo[i] ? wg_octo : wg_launch) * work[i] / total[octo[i] ? 1 : 0] + 0.5);
    if (budget < P.pairs) budget = P.pairs;
    const int rc = wgrad_plan_one<T, NT>(descs[i], P, budget, octo[i]);
    if (rc != HRP_OK) return rc;
    const int64_t need = (int64_t)P.t.G * P.fold_pairs * P.nte * 1024 * 4;
    info->ws_bytes[i] = need;
    if (tab) {
      HRP_REQUIRE(descs[i].workspace && descs[i].workspace_bytes >= need,
                  "wgrad batch: problem %d needs %lld workspace bytes (has %lld)", i, (long long)need, (long long)descs[i].workspace_bytes);
      HRP_REQUIRE((uintptr_t)descs[i].workspace % 16 == 0, "wgrad batch: workspace alignment");
    }
    P.t.use_ws = 1;
    info->blk0[i] = blk;
    if (octo[i]) {
      lds_octo = P.t.lds_bytes > lds_octo ? P.t.lds_bytes : lds_octo;
      P.wblk0 = blkw;
      blkw += P.t.G * P.pairs;
    } else {
      lds_max = P.t.lds_bytes > lds_max ? P.t.lds_bytes : lds_max;
      blk += P.t.G * P.pairs;
    }
    info->blk2[i] = blk2;
    blk2 += P.nte * 4 * P.fold_pairs;
    if (tab) tab[i] = P;
  }
  info->blk0[n] = blk; info->blk2[n] = blk2;
  // phase 1 (all problems or none): the caller folds the slabs with a HRP_BATCH_WGRAD_FOLD launch of its own
  info->grid = blk; info->grid2 = descs[0].phase == 1 ? 0 : blk2;
  info->grid3 = blkw; info->lds_bytes3 = lds_octo;
  info->lds_bytes = lds_max;
  info->variant = NT;
  return HRP_OK;
}

template <typename T>
static int wgrad_batch_prepare_t(const hrp_wgrad_desc* descs, int n, void* table, hrp_batch_info* info) {
  WgradProblem* tab = (WgradProblem*)table;
  switch (descs[0].ntaps) {
    case 1: return wgrad_batch_prepare_nt<T, 1>(descs, n, tab, info);
    case 4: return wgrad_batch_prepare_nt<T, 4>(descs, n, tab, info);
    default: return wgrad_batch_prepare_nt<T, 9>(descs, n, tab, info);
  }
}

int wgrad_batch_prepare(const hrp_wgrad_desc* descs, int n, void* table, hrp_batch_info* info) {
  for (int i = 0; i < n; ++i) {
    const int rc = wgrad_check(&descs[i]);
    if (rc != HRP_OK) return rc;
    HRP_REQUIRE(descs[i].ntaps == descs[0].ntaps && descs[i].dtype == descs[0].dtype, "wgrad batch: mixed tap counts / element types");
    HRP_REQUIRE(descs[i].phase == descs[0].phase, "wgrad batch: mixed phases");
  }
  if (descs[0].dtype == HRP_F32) return wgrad_batch_prepare_t<float>(descs, n, table, info);
  if (descs[0].dtype == HRP_F32X3) return wgrad_batch_prepare_t<f32x3_t>(descs, n, table, info);
  return wgrad_batch_prepare_t<bf16_t>(descs, n, table, info);
}

template <typename T, int NT>
static int wgrad_batch_launch_nt(const WgradProblem* tab, const hrp_batch_info* info, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_batch_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  int rc = HRP_OK;
  if (info->grid > 0) {
    hipLaunchKernelGGL((wgrad_batch_kernel<T, NT>), dim3(info->grid), dim3(256), info->lds_bytes, s, tab, make_hdr(info->blk0, info->n));
    rc = check_launch("wgrad_batch_kernel");
  }
  if (rc == HRP_OK && info->grid3 > 0) {
    void (*okern)(const WgradProblem*, const int) = std::is_same<T, f32x3_t>::value ? wgrad_octo_x3_batch_kernel<(NT == 1 ? 1 : 9)> : wgrad_octo_batch_kernel;
    static bool octo_attr = false;
    if (!octo_attr) {
      (void)hipFuncSetAttribute((const void*)okern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      octo_attr = true;
    }
    hipLaunchKernelGGL(okern, dim3(info->grid3), dim3(512), info->lds_bytes3, s, tab, info->n);
    rc = check_launch("wgrad_octo_batch_kernel");
  }
  if (rc != HRP_OK || info->grid2 == 0) return rc;
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(info->grid2), dim3(256), 0, s, tab, make_hdr(info->blk2, info->n));
  return check_launch("wgrad_reduce_batch_kernel");
}

int wgrad_batch_launch(const void* table_dev, const hrp_batch_info* info, hipStream_t s) {
  const WgradProblem* tab = (const WgradProblem*)table_dev;
  if (info->dtype == HRP_F32) {
    switch (info->variant) {
      case 1: return wgrad_batch_launch_nt<float, 1>(tab, info, s);
      case 4: return wgrad_batch_launch_nt<float, 4>(tab, info, s);
      case 9: return wgrad_batch_launch_nt<float, 9>(tab, info, s);
    }
  } else if (info->dtype == HRP_F32X3) {
    switch (info->variant) {
      case 1: return wgrad_batch_launch_nt<f32x3_t, 1>(tab, info, s);
      case 4: return wgrad_batch_launch_nt<f32x3_t, 4>(tab, info, s);
      case 9: return wgrad_batch_launch_nt<f32x3_t, 9>(tab, info, s);
    }
  } else {
    switch (info->variant) {
      case 1: return wgrad_batch_launch_nt<bf16_t, 1>(tab, info, s);
      case 4: return wgrad_batch_launch_nt<bf16_t, 4>(tab, info, s);
      case 9: return wgrad_batch_launch_nt<bf16_t, 9>(tab, info, s);
    }
  }
  set_error("wgrad batch: bad variant %d", info->variant);
  return HRP_ERR_ARG;
}

int64_t wgrad_batch_table_bytes(int n) { return (int64_t)n * sizeof(WgradProblem); }

// ---- deferred folds ---------------------------------------------------------------------------------------------
int64_t wgrad_fold_table_bytes(int n) { return (int64_t)n * sizeof(FoldProblem); }

int wgrad_fold_prepare(const hrp_wgrad_fold_desc* descs, int n, void* table, hrp_batch_info* info) {
  FoldProblem* tab = (FoldProblem*)table;
  int blk = 0;
  for (int i = 0; i < n; ++i) {
    const hrp_wgrad_fold_desc& f = descs[i];
    HRP_REQUIRE(f.workspace && f.dw && f.G >= 1 && f.pairs >= 1 && f.n_cib >= 1 && f.pairs % f.n_cib == 0, "wgrad fold: problem %d: bad descriptor", i);
    HRP_REQUIRE((f.nb == 1 || f.nb == 2) && f.nte >= 1 && f.nte % (f.nb * f.nb) == 0 && f.nte / (f.nb * f.nb) == f.ntaps, "wgrad fold: problem %d: tile elements", i);
    HRP_REQUIRE(f.dw_tap_stride == 0 || (f.dw_tap_off >= 0 && f.dw_tap_off + f.ntaps <= f.dw_tap_stride), "wgrad fold: tap group");
    info->blk0[i] = blk;
    blk += f.nte * 4 * f.pairs;
    if (tab) {
      memset(&tab[i], 0, sizeof(FoldProblem));
      tab[i].f = f;
      tab[i].fd_r = make_fastdiv(f.nte * 4);
    }
  }
  info->blk0[n] = blk;
  info->grid = blk; info->grid2 = 0; info->lds_bytes = 0; info->variant = 0; info->dtype = HRP_F32;
  return HRP_OK;
}

int wgrad_fold_launch(const void* table_dev, const hrp_batch_info* info, hipStream_t s) {
  hipLaunchKernelGGL(wgrad_fold_batch_kernel, dim3(info->grid), dim3(256), 0, s, (const FoldProblem*)table_dev, make_hdr(info->blk0, info->n));
  return check_launch("wgrad_fold_batch_kernel");
}

template <typename T, int NT, int NB>
static int fold_desc_nb(const hrp_wgrad_desc& d, hrp_wgrad_fold_desc* out) {
  WgradTiling t{};
  const int rc = wgrad_tiling<T, NT, NB>(d, t);
  if (rc != HRP_OK) return rc;
  const int pairs = t.n_cob * t.n_cib;
  const int64_t need = (int64_t)t.G * pairs * (NT * NB * NB) * 1024 * 4;
  const bool use_ws = d.workspace && d.workspace_bytes >= need;   // (as launch_wgrad_nb decides)
  *out = make_fold_desc(d, use_ws ? t.G : 0, pairs, t.n_cib, NT * NB * NB, NB);
  return HRP_OK;
}
template <typename T, int NT>
static int fold_desc_t(const hrp_wgrad_desc& d, hrp_wgrad_fold_desc* out) {
  if constexpr (can_nb2<T, NT>()) {
    if (want_nb2(d)) return fold_desc_nb<T, NT, 2>(d, out);
  }
  return fold_desc_nb<T, NT, 1>(d, out);
}

}  // namespace hrp

extern "C" int hrp_wgrad_fold_desc_of(const hrp_wgrad_desc* d, hrp_wgrad_fold_desc* out) {
  using namespace hrp;
  HRP_REQUIRE(out, "wgrad fold: null pointer");
  const int crc = wgrad_check(d);
  if (crc != HRP_OK) return crc;
  if (d->dtype == HRP_F32) {
    if (d->ntaps == 1) return fold_desc_t<float, 1>(*d, out);
    if (d->ntaps == 4) return fold_desc_t<float, 4>(*d, out);
    return fold_desc_t<float, 9>(*d, out);
  }
  if (d->dtype == HRP_F32X3) {
    if (d->ntaps == 1) return fold_desc_t<f32x3_t, 1>(*d, out);
    if (d->ntaps == 4) return fold_desc_t<f32x3_t, 4>(*d, out);
    return fold_desc_t<f32x3_t, 9>(*d, out);
  }
  if (d->ntaps == 1) return fold_desc_t<bf16_t, 1>(*d, out);
  if (d->ntaps == 4) return fold_desc_t<bf16_t, 4>(*d, out);
  return fold_desc_t<bf16_t, 9>(*d, out);
}

extern "C" int hrp_batch_wgrad_fold_descs(const void* table_host, const hrp_batch_info* info, hrp_wgrad_fold_desc* out) {
  using namespace hrp;
  HRP_REQUIRE(table_host && info && out && info->family == HRP_BATCH_WGRAD, "wgrad fold: needs a prepared HRP_BATCH_WGRAD table");
  const WgradProblem* tab = (const WgradProblem*)table_host;
  for (int i = 0; i < info->n; ++i)
    out[i] = make_fold_desc(tab[i].d, tab[i].t.G, tab[i].fold_pairs, tab[i].fold_n_cib, tab[i].nte, tab[i].fold_nb);
  return HRP_OK;
}

extern "C" int hrp_conv2d_bwd_weight(const hrp_wgrad_desc* d, void* stream) {
  using namespace hrp;
  const int crc = wgrad_check(d);
  if (crc != HRP_OK) return crc;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == HRP_F32) {
    if (d->ntaps == 1) return launch_wgrad<float, 1>(*d, s);
    if (d->ntaps == 4) return launch_wgrad<float, 4>(*d, s);
    return launch_wgrad<float, 9>(*d, s);
  }
  if (d->dtype == HRP_F32X3) {
    if (d->ntaps == 1) return launch_wgrad<f32x3_t, 1>(*d, s);
    if (d->ntaps == 4) return launch_wgrad<f32x3_t, 4>(*d, s);
    return launch_wgrad<f32x3_t, 9>(*d, s);
  }
  if (d->ntaps == 1) return launch_wgrad<bf16_t, 1>(*d, s);
  if (d->ntaps == 4) return launch_wgrad<bf16_t, 4>(*d, s);
  return launch_wgrad<bf16_t, 9>(*d, s);
}

extern "C" int64_t hrp_wgrad_workspace_bytes(const hrp_wgrad_desc* d) {
  using namespace hrp;
  if (!d) return 0;
  if (d->dtype == HRP_F32) {
    if (d->ntaps == 1) return wgrad_ws_bytes<float, 1>(*d);
    if (d->ntaps == 4) return wgrad_ws_bytes<float, 4>(*d);
    return wgrad_ws_bytes<float, 9>(*d);
  }
  if (d->dtype == HRP_F32X3) {
    if (d->ntaps == 1) return wgrad_ws_bytes<f32x3_t, 1>(*d);
    if (d->ntaps == 4) return wgrad_ws_bytes<f32x3_t, 4>(*d);
    return wgrad_ws_bytes<f32x3_t, 9>(*d);
  }
  if (d->ntaps == 1) return wgrad_ws_bytes<bf16_t, 1>(*d);
  if (d->ntaps == 4) return wgrad_ws_bytes<bf16_t, 4>(*d);
  return wgrad_ws_bytes<bf16_t, 9>(*d);
}
