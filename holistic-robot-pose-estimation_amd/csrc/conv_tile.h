// Tile convolution on MFMA for gfx950: one kernel family for conv k1/k3, stride 1/2, its data gradient
// and nn.Linear (see hrp_conv_desc in include/hrp.h).
//
// Work decomposition
//   workgroup (256 threads = 4 waves) -> output tile of BM = TI*TH*TW pixels x BN output channels
//   K loop = input-channel chunks of 32 bytes (16 bf16 / 8 fp32) x taps x MFMA k-steps
//   LDS   = two stage buffers, each: input halo tile [pixel][32 B] (read once per chunk, reused by every
//           tap) + weight slab [tap][cout][32 B]
//   staging = direct-to-LDS DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR round trip):
//           chunk c+1 is in flight while chunk c is multiplied; one barrier per chunk.  The DMA writes
//           lane-linear, so the 16-byte XOR swizzle that keeps ds_read_b128 conflict-free is applied on the
//           per-lane SOURCE address (row r keeps its two 16-byte halves swapped when bit 3 of r is set) and
//           again on the read.  Out-of-image pixels (zero padding) read a 64-byte zero page instead.
//   MFMA  = 32x32x16 bf16 / 32x32x2 fp32, A = weights (rows = cout), B = pixels (cols = pixel):
//           each lane ends up with 4 consecutive output channels of one pixel per accumulator quad,
//           so the epilogue moves 8/16-byte pieces through LDS and leaves as 16-byte coalesced stores.
//   epilogue = bias, per-channel affine (folded BN), residual add, ReLU, per-channel sum / sum-of-squares
//           (train-mode BN statistics, one atomicAdd per channel per workgroup into one of 8 slots).
//   block id -> XCD-contiguous remap so the cout blocks / neighbouring tiles that share an input tile hit
//           the same XCD's L2.
#pragma once
#include "hrp_common.h"
#include "batch.h"
#include <stdlib.h>
#include <type_traits>

#ifndef HRP_CONV_ISSUE_STEPS
#define HRP_CONV_ISSUE_STEPS 64
#endif

namespace hrp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ROW = 32;  // bytes of input channels per pixel / weight row staged per K chunk

static __device__ uint4 g_zero_page[1024];  // 16 KiB of zeros: DMA source of padding pixels / rows beyond the tensor (a lane parked
                                     // here still advances 32 bytes per K chunk, <= 8.3 KiB for the widest layer)

struct ConvTiling {
  int TH, TW, TI;   // TI = images per tile actually staged (TI*TH*TW <= BM; the rest of the tile is idle)
  int IHt, IWt;
  int mindy, mindx;
  int tiles_x, tiles_y, tiles_n;
  int n_cout_blk;
  int in_rows;              // TI*IHt*IWt
  int in_pieces, w_pieces;  // 1 KiB DMA pieces per stage
  int buf_bytes;            // one 32-byte-chunk buffer (input rows + weight rows)
  int G;                    // chunks per pipeline stage
  int lds_stats_off;
  int nblocks;
  int vec_ok;
  int ksplit, cps;           // split-K (fp32 skinny layers): K slices per output tile, chunks per slice
  FastDiv fd_ihw, fd_iwt, fd_thw, fd_tw, fd_ncb, fd_tx, fd_ty;
};

// byte offset of (row r, 16-byte half h) inside a staged region
__device__ __forceinline__ int row_addr(int r, int h) { return r * ROW + ((h ^ ((r >> 3) & 1)) << 4); }

// -DHRP_TIMELINE (development build only): thread 0 of every workgroup stamps the 100 MHz wall clock at
// phase boundaries into g_conv_timeline[block][8]; tools/bench_kernels.py reads it with hrp_debug_conv_timeline.
#ifdef HRP_TIMELINE
static __device__ unsigned long long g_conv_timeline[8192 * 8];
#define HRP_CSTAMP(i) do { if (tid == 0 && blockIdx.x < 8192) g_conv_timeline[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HRP_CSTAMP(i) do { } while (0)
#endif

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  using Frag = bf16x8;
  // one k-step per 32-byte row: lanes 0-31 take the first 16 bytes (k 0..7), lanes 32-63 the second
  static constexpr int KSTEPS = 1;
  __device__ static __forceinline__ Frag ld(const char* base, int r, int kk, int khalf) {
    return *(const Frag*)(base + row_addr(r, khalf));
  }
  __device__ static __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  using Frag = float;
  static constexpr int KSTEPS = 4;  // 8 fp32 per row, 2 per MFMA
  __device__ static __forceinline__ Frag ld(const char* base, int r, int kk, int khalf) {
    return *(const float*)(base + row_addr(r, kk >> 1) + (kk & 1) * 8 + khalf * 4);
  }
  __device__ static __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma<f32x3_t> {      // (the MFMA loop of this type is its own branch of conv_tile_body; KSTEPS feeds the host heuristics only)
  using Frag = float;
  static constexpr int KSTEPS = 4;
};

// x rotated right by N lanes inside each row of 16 lanes (DPP row_ror)
template <int N>
__device__ __forceinline__ float dpp_row_ror(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}

// 64 lanes x 16 bytes global -> LDS (lane-linear at lds_wave_base).  Inline assembly on purpose: behind the
// builtin the compiler's wait-count pass may drain the DMA (s_waitcnt vmcnt(0)) before LDS reads it cannot
// prove disjoint, which would serialise the prefetch with the MFMAs.  The kernel waits explicitly per stage.
__device__ __forceinline__ void dma16(const char* src, char* lds_wave_base) {
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :: "v"(src), "s"(lds) : "memory");   // m0 is scratch for the compiler too: it never keeps a value there
}

// The whole tile program.  blk0 / blk_stride: the block of this problem the workgroup starts at and (PERSIST) its
// stride; stat_slot: which of the HRP_STAT_SLOTS statistic replicas this workgroup adds into.  Called by the
// single-problem kernel below with (blockIdx.x, gridDim.x) and by the batched kernel (conv_batch.h) with the block
// index inside the problem it looked up.
template <typename T, int CT, int PT, int WC, int WP, int NT, bool PERSIST, bool FAST>
__device__ __forceinline__ void conv_tile_body(const hrp_conv_desc& d, const ConvTiling& t, const int blk0,
                                               const int blk_stride, const int stat_slot) {
  static_assert(WC * WP == 4, "4 waves");
  constexpr int BN = 32 * CT * WC, BM = 32 * PT * WP;
  constexpr int SZ = Elem<T>::SZ, VEC = Elem<T>::VEC;
  constexpr int CKE = ROW / SZ;  // channels per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WC, wp = wave / WC;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int IS = d.in_stride;
  const int thw = t.TH * t.TW;
  const int ihw = t.IHt * t.IWt;
  HRP_CSTAMP(0);

  // =====================================================================================================
  // Tile independent state, computed once.  PERSIST: the workgroup walks tiles b, b + grid, ... (for small
  // layers the index arithmetic below costs more instructions than the MFMA loop of a tile); otherwise one
  // tile per workgroup, which keeps the register count - and with it the occupancy of deep-K layers - low.
  // =====================================================================================================
  int pixrow[PT];   // filled by late_setup()
  const int wrow0 = wc * 32 * CT + l31;

  const int nchunks = (d.Cin + CKE - 1) / CKE;
  const char* xg = (const char*)d.x;
  const char* wg = (const char*)d.w;
  const char* zero_sym = (const char*)g_zero_page;
  const char* zero = zero_sym;
  // keep the zero-page pointer in a VGPR pair: left to itself the compiler rematerialises it (s_getpc, 2 s_add,
  // 2 v_mov) in front of every DMA piece, a third of the instructions of the K loop
  asm volatile("" : "+v"(zero));
  // tap offsets (in tile rows) live in registers: with NT known the tap loop unrolls completely
  int taprow[NT];
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) taprow[tp] = (d.dy[tp] - t.mindy) * t.IWt + (d.dx[tp] - t.mindx);

  // ---- DMA plan of this wave -------------------------------------------------------------------------
  // Per 1 KiB input piece a lane keeps the byte offset of its 16-byte slot relative to the tile origin
  // (chunk 0) and a validity code: bit 0/1 = outside the image in the first / last tile row, bit 2/3 = same
  // for tile columns, bit 4 = beyond the batch in the last image group, bit 5 = never fetched (padding of
  // the staged region), bit 6 = second half of a chunk (invalid in a half-filled last chunk).  A tile's
  // class mask selects the bits that apply; the host checks that only border tiles can leave the image.
  constexpr int MAXP_IN = 10;                    // input pieces per wave (tile <= 40 KiB per chunk)
  constexpr int W_PIECES = NT * BN / 32;         // weight pieces per chunk (BN % 32 == 0)
  constexpr int MAXP_W = (W_PIECES + 3) / 4;
  int in_rel[MAXP_IN], in_code[MAXP_IN];
  const int y_last = (t.tiles_y - 1) * t.TH, x_last = (t.tiles_x - 1) * t.TW, n_last = (t.tiles_n - 1) * t.TI;
  {
    const int iy_last = y_last * IS + t.mindy, ix_last = x_last * IS + t.mindx;
#pragma unroll
    for (int i = 0; i < MAXP_IN; ++i) {
      in_rel[i] = 0; in_code[i] = 32;
      if (wave + 4 * i >= t.in_pieces) continue;
      const int sl = (wave + 4 * i) * 64 + lane;   // 16-byte slot of the input region
      const int r = sl >> 1;                       // tile pixel row
      const int h = (sl & 1) ^ ((r >> 3) & 1);     // logical half stored in this slot
      const int ti = fdiv16(r, t.fd_ihw), rem = r - mul24(ti, ihw);
      const int iy = fdiv16(rem, t.fd_iwt), ix = rem - mul24(iy, t.IWt);
      int code = (r >= t.in_rows) ? 32 : 0;
      code |= (iy + t.mindy < 0) ? 1 : 0;
      code |= (iy + iy_last >= d.H) ? 2 : 0;
      code |= (ix + t.mindx < 0) ? 4 : 0;
      code |= (ix + ix_last >= d.W) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      code |= h ? 64 : 0;
      in_code[i] = code;
      in_rel[i] = mul24(mul24(mul24(ti, d.H) + iy, d.W) + ix, d.x_pitch * SZ) + h * (VEC * SZ);
    }
  }
  int w_rel[MAXP_W];                             // byte offset into the packed weights (chunk 0, cout block 0)
#pragma unroll
  for (int i = 0; i < MAXP_W; ++i) {
    w_rel[i] = 0;
    const int p = wave + 4 * i;
    if (p < W_PIECES) {
      const int sl = p * 64 + lane;
      const int r = sl >> 1;                       // tap-major weight row: tl * BN + j
      const int h = (sl & 1) ^ ((r >> 3) & 1);
      const int tl = (p * 32) / BN;                // a piece is 32 rows and BN is a multiple of 32: one tap
      const int j = r - tl * BN;
      w_rel[i] = (d.wtap[tl] * d.w_cout_pad + j) * ROW + h * 16;
    }
  }
  const int w_chunk_stride = d.w_ntaps * d.w_cout_pad * ROW;  // bytes between consecutive chunks of the packing
  const int half_chunk = (d.Cin % CKE) ? nchunks - 1 : -1;    // chunk whose second 16 bytes lie beyond Cin

  // ---- store plan: the rows of the output tile this thread writes in the coalesced pass ----------------
  constexpr int NV = BN / VEC, KST = BM * NV / 256;
  const int cv = tid % NV;  // 256 % NV == 0, so a thread keeps its channel group
  int out_rel[KST], out_code[KST];
  // The part of the setup that the first DMA stage does not need (fragment rows, store plan) runs after that stage
  // has been issued, under its latency.
  auto late_setup = [&]() {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      int m = wp * (32 * PT) + pt * 32 + l31;
      int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      if (ti >= t.TI) ti = t.TI - 1;  // idle slot of a partially filled tile: read something valid, never stored
      pixrow[pt] = mul24(mul24(ti, t.IHt) + mul24(ty, IS), t.IWt) + mul24(tx, IS);
    }
#pragma unroll
    for (int k = 0; k < KST; ++k) {
      const int m = tid / NV + k * (256 / NV);
      const int ti = fdiv16(m, t.fd_thw), rem = m - mul24(ti, thw);
      const int ty = fdiv16(rem, t.fd_tw), tx = rem - mul24(ty, t.TW);
      int code = (ti >= t.TI) ? 32 : 0;
      code |= (ty + y_last >= d.Ho) ? 2 : 0;
      code |= (tx + x_last >= d.Wo) ? 8 : 0;
      code |= (ti + n_last >= d.N) ? 16 : 0;
      out_code[k] = code;
      out_rel[k] = mul24(mul24(ti, d.y_H) + mul24(ty, d.out_stride), d.y_W) + mul24(tx, d.out_stride);   // in output pixels
    }
  };
  bool first_tile = true;
  HRP_CSTAMP(1);

  const int G = t.G, stage_bytes = G * t.buf_bytes;
  constexpr int OP = BN * SZ + 16;
  char* lds_out = smem;
  float* lds_stats = (float*)(smem + t.lds_stats_off);

  for (int b = blk0; PERSIST ? b < t.nblocks : b == blk0; b += PERSIST ? blk_stride : 1) {
    int bid = b;
    if ((t.nblocks & 7) == 0) bid = (bid & 7) * (t.nblocks >> 3) + (bid >> 3);
    int kz = 0;
    if (t.ksplit > 1) { kz = bid % t.ksplit; bid /= t.ksplit; }
    const int cbeg = kz * t.cps;                                     // first chunk of this K slice
    const int nloc = t.ksplit > 1 ? min(t.cps, nchunks - cbeg) : nchunks;
    const int nstages = (nloc + G - 1) / G;
    const int tile = fdiv(bid, t.fd_ncb);
    const int cb = bid - tile * t.n_cout_blk;
    const int q = fdiv(tile, t.fd_tx);
    const int tx_i = tile - q * t.tiles_x;
    const int tn_i = fdiv(q, t.fd_ty);
    const int ty_i = q - tn_i * t.tiles_y;
    const int n0 = tn_i * t.TI, oy0 = ty_i * t.TH, ox0 = tx_i * t.TW;
    const int iy0 = oy0 * IS + t.mindy, ix0 = ox0 * IS + t.mindx;
    const int co0 = cb * BN;
    const int cls = 32 | (ty_i == 0 ? 1 : 0) | (ty_i == t.tiles_y - 1 ? 2 : 0) | (tx_i == 0 ? 4 : 0) |
                    (tx_i == t.tiles_x - 1 ? 8 : 0) | (tn_i == t.tiles_n - 1 ? 16 : 0);
    // tile origins (the input origin may lie before the tensor: only valid lanes dereference it)
    const char* xbase = xg + (((long long)n0 * d.H + iy0) * d.W + ix0) * (long long)d.x_pitch * SZ;
    const char* wbase = wg + (long long)co0 * ROW;

    f32x16 acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][p][i] = 0.f;

    // Per-piece source pointers of this tile, advanced by one chunk after every issue (chunks are issued in
    // order, exactly once): the K loop then pays one 64-bit add per piece instead of mask / compare / select.
    // Not with a half-filled last chunk (Cin % 16 != 0: its upper 16 bytes must read zeros) - that keeps the
    // select path.
    constexpr bool fast = FAST;    // host: Cin * sizeof(T) is a multiple of 32
    const char* pin[MAXP_IN];
    const char* pw[MAXP_W];
    int wstep[MAXP_W];
    if constexpr (fast) {
      const char* xb0 = xbase + (long long)cbeg * ROW;
#pragma unroll
      for (int i = 0; i < MAXP_IN; ++i) pin[i] = (in_code[i] & cls) ? zero : xb0 + (unsigned)in_rel[i];
      const char* wb0 = wbase + (long long)cbeg * w_chunk_stride;
#pragma unroll
      for (int i = 0; i < MAXP_W; ++i) {
        const bool ok = co0 + ((wave + 4 * i) * 32) % BN < d.w_cout_pad;
        pw[i] = ok ? wb0 + (unsigned)w_rel[i] : zero;
        wstep[i] = ok ? w_chunk_stride : 0;
      }
    }
    // DMA slot = one 1 KiB piece of this wave: slots 0 .. MAXP_IN-1 belong to the input tile of the chunk,
    // the rest to its weight slab (slot is a constant after unrolling)
    auto issue_slot = [&](int chunk, char* buf, int slot) {
      if constexpr (fast) {
        if (slot < MAXP_IN) {
          const int p = wave + 4 * slot;
          if (p < t.in_pieces) {
            dma16(pin[slot], buf + p * 1024);
            pin[slot] += ROW;
          }
        } else if (slot < MAXP_IN + MAXP_W) {
          const int i = slot - MAXP_IN, p = wave + 4 * i;
          if (p < W_PIECES) {
            dma16(pw[i], buf + (t.in_pieces + p) * 1024);
            pw[i] += wstep[i];
          }
        }
        return;
      }
      if (slot < MAXP_IN) {
        const int p = wave + 4 * slot;
        if (p < t.in_pieces) {
          const int ccls = cls | (chunk == half_chunk ? 64 : 0);
          dma16((in_code[slot] & ccls) ? zero : xbase + (long long)chunk * ROW + (unsigned)in_rel[slot], buf + p * 1024);
        }
      } else if (slot < MAXP_IN + MAXP_W) {
        const int i = slot - MAXP_IN, p = wave + 4 * i;
        if (p < W_PIECES) {
          const bool ok = co0 + (p * 32) % BN < d.w_cout_pad;   // a piece = 32 cout rows of one tap: uniform
          dma16(ok ? wbase + (long long)chunk * w_chunk_stride + (unsigned)w_rel[i] : zero,
                buf + (t.in_pieces + p) * 1024);
        }
      }
    };
    // A stage = G consecutive 32-byte chunks (G sub-buffers): the DMA latency of a stage is paid once per G
    // chunks of MFMA work instead of once per chunk (deep-K layers have 16 chunks of only 9 MFMAs per wave).
    auto issue_stage = [&](int st, char* base) {
      for (int g = 0; g < G; ++g)
        if (st * G + g < nloc) {
#pragma unroll
          for (int slot = 0; slot < MAXP_IN + MAXP_W; ++slot) issue_slot(cbeg + st * G + g, base + g * t.buf_bytes, slot);
        }
    };
    issue_stage(0, smem);
    if (first_tile) { late_setup(); first_tile = false; }
    HRP_CSTAMP(2);
    for (int st = 0; st < nstages; ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of stage `st` have landed
      __syncthreads();                                   // everyone's have, and stage st-1 has been consumed
      if (st == 0) HRP_CSTAMP(3);
      char* sbuf = smem + (st & 1) * stage_bytes;
      char* nbuf = smem + ((st + 1) & 1) * stage_bytes;
      const bool more = st + 1 < nstages;
      // One step = one (tap, k-step) of a chunk: CT weight fragments, PT pixel fragments, CT x PT MFMAs.  The
      // fragments of step s+1 (also across the chunk boundary inside the stage) are read while the MFMAs of
      // step s run: with one or two waves per SIMD nothing else would hide the LDS latency.
      // The pieces of stage st+1 are issued between the MFMA steps (chunk g of the next stage during chunk g of
      // this one, SPP slots per step): issued in one burst a full memory queue stalls the wave for ~2 us.
      const int ng = nloc - st * G < G ? nloc - st * G : G;
      if constexpr (std::is_same<T, f32x3_t>::value) {
        // ---- fp32 tensors, 3 x bf16 products.  One v_mfma_f32_32x32x16_bf16 takes 16 K values, 8 per lane half.  Weight rows
        // arrive PRE-SPLIT from hrp_pack_weights (logical half 0 = 8 hi, half 1 = 8 lo); the pixels' 8 floats are split here (24
        // VALU per fragment, under the MFMAs it feeds).  Two forms: a chunk PAIR (1x1 layers, whose stages hold several 8-channel
        // chunks: lane half 0 works on chunk g, half 1 on chunk g + 1, three MFMAs per 16 channels) and a SINGLE chunk (the 3x3
        // layers, whose stages hold one chunk: two MFMAs per 8 channels, below).  The fp32 MFMA path spends 8 CT PT instructions of
        // twice the latency on the products of one chunk.
        constexpr int SPX = (MAXP_IN + MAXP_W + NT - 1) / NT;
        const int khoff = khalf * t.buf_bytes;
        auto ldpix = [&](const char* lds_in, int tap, float4 (&b0)[PT], float4 (&b1)[PT]) {
#pragma unroll
          for (int p = 0; p < PT; ++p) {
            const int rr = pixrow[p] + taprow[tap];
            b0[p] = *(const float4*)(lds_in + row_addr(rr, 0));
            b1[p] = *(const float4*)(lds_in + row_addr(rr, 1));
          }
        };
        auto issue_next = [&](int gv, int tap, bool both) {      // the pieces of chunk g (and g + 1) of the next stage
          const int g = __builtin_amdgcn_readfirstlane(gv);      // (uniform; the DMA's LDS base must sit in an SGPR)
          if (more) {
#pragma unroll
            for (int u = 0; u < SPX; ++u) {
              if ((st + 1) * G + g < nloc) issue_slot(cbeg + (st + 1) * G + g, nbuf + g * t.buf_bytes, tap * SPX + u);
              if (both && (st + 1) * G + g + 1 < nloc && g + 1 < G) issue_slot(cbeg + (st + 1) * G + g + 1, nbuf + (g + 1) * t.buf_bytes, tap * SPX + u);
            }
          }
        };
        // (3x3 layers: every chunk on its own - their stages hold one chunk anyway, and the pair form next to the single form in a
        // nine- or four-tap kernel spills; the 1x1 layers, whose stages hold several chunks, pair them)
        constexpr bool PAIRS = NT <= 2;
        for (int g = 0; g < ng; g += PAIRS ? 2 : 1) {
          if (PAIRS && g + 1 < ng) {
            // ---- a chunk pair: lane half 0 works on chunk g, half 1 on chunk g + 1; lo*hi, hi*lo, hi*hi per (cout, pixel) tile
            const char* lds_in = sbuf + g * t.buf_bytes + khoff;
            const char* lds_w = lds_in + t.in_pieces * 1024;
            struct Raw { uint4 ah[CT], al[CT]; float4 b0[PT], b1[PT]; };
            auto ldraw = [&](int tap, Raw& r) {
              const int wr = tap * BN + wrow0;
#pragma unroll
              for (int c = 0; c < CT; ++c) { r.ah[c] = *(const uint4*)(lds_w + row_addr(wr + c * 32, 0)); r.al[c] = *(const uint4*)(lds_w + row_addr(wr + c * 32, 1)); }
              ldpix(lds_in, tap, r.b0, r.b1);
            };
            Raw cur, nxt;
            ldraw(0, cur);
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) {
              if (tap + 1 < NT) ldraw(tap + 1, nxt);
              uint4 bh[PT], bl[PT];
#pragma unroll
              for (int p = 0; p < PT; ++p) {
                const float x[8] = {cur.b0[p].x, cur.b0[p].y, cur.b0[p].z, cur.b0[p].w, cur.b1[p].x, cur.b1[p].y, cur.b1[p].z, cur.b1[p].w};
                split_bf16x8(x, bh[p], bl[p]);
              }
#pragma unroll
              for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                  const bf16x8 ah = __builtin_bit_cast(bf16x8, cur.ah[c]), al = __builtin_bit_cast(bf16x8, cur.al[c]);
                  const bf16x8 xh = __builtin_bit_cast(bf16x8, bh[p]), xl = __builtin_bit_cast(bf16x8, bl[p]);
                  acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, acc[c][p], 0, 0, 0);      // (small terms first)
                  acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, acc[c][p], 0, 0, 0);
                  acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, acc[c][p], 0, 0, 0);
                }
              issue_next(g, tap, true);
              if (tap + 1 < NT) cur = nxt;
            }
          } else {
            // ---- a chunk without a partner (G = 1 - the usual case: fp32 tiles and hi | lo weight rows leave room for one chunk
            // per stage - or the last of an odd count): the MFMA's K is [w_hi | w_lo] of THIS chunk (one 16-byte read per lane
            // half), against [x_lo | 0] = w_hi x_lo and against [x_hi | x_hi] = w_hi x_hi + w_lo x_hi: TWO MFMAs per tile (a
            // zero-padded partner chunk took three, half of each idle)
            const char* lds_in = sbuf + g * t.buf_bytes;
            const char* lds_w = lds_in + t.in_pieces * 1024;
            struct Raw1 { uint4 a[CT]; float4 b0[PT], b1[PT]; };
            auto ldraw1 = [&](int tap, Raw1& r) {
              const int wr = tap * BN + wrow0;
#pragma unroll
              for (int c = 0; c < CT; ++c) r.a[c] = *(const uint4*)(lds_w + row_addr(wr + c * 32, khalf));
              ldpix(lds_in, tap, r.b0, r.b1);
            };
            Raw1 cur, nxt;
            ldraw1(0, cur);
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) {
              if (tap + 1 < NT) ldraw1(tap + 1, nxt);
              uint4 bh[PT], bl[PT];
#pragma unroll
              for (int p = 0; p < PT; ++p) {
                const float x[8] = {cur.b0[p].x, cur.b0[p].y, cur.b0[p].z, cur.b0[p].w, cur.b1[p].x, cur.b1[p].y, cur.b1[p].z, cur.b1[p].w};
                split_bf16x8(x, bh[p], bl[p]);
                if (khalf) bl[p] = make_uint4(0, 0, 0, 0);
              }
#pragma unroll
              for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                  const bf16x8 a = __builtin_bit_cast(bf16x8, cur.a[c]);
                  const bf16x8 xh = __builtin_bit_cast(bf16x8, bh[p]), xl = __builtin_bit_cast(bf16x8, bl[p]);
                  acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xl, acc[c][p], 0, 0, 0);
                  acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xh, acc[c][p], 0, 0, 0);
                }
              issue_next(g, tap, PAIRS);
              if (tap + 1 < NT) cur = nxt;
            }
          }
        }
      } else {
      constexpr int NS = NT * Mma<T>::KSTEPS;
      constexpr int ISSUE_STEPS = NS < HRP_CONV_ISSUE_STEPS ? NS : HRP_CONV_ISSUE_STEPS;   // front-load the next stage's DMA
      constexpr int SPP = (MAXP_IN + MAXP_W + ISSUE_STEPS - 1) / ISSUE_STEPS;
      typename Mma<T>::Frag fa[2][CT], fb[2][PT];
      auto load = [&](const char* lds_in, int step, typename Mma<T>::Frag (&a)[CT], typename Mma<T>::Frag (&bb)[PT]) {
        const int tap = step / Mma<T>::KSTEPS, kk = step % Mma<T>::KSTEPS;   // constants after unrolling
        const char* lds_w = lds_in + t.in_pieces * 1024;
        const int wr = tap * BN + wrow0;
#pragma unroll
        for (int c = 0; c < CT; ++c) a[c] = Mma<T>::ld(lds_w, wr + c * 32, kk, khalf);
#pragma unroll
        for (int p = 0; p < PT; ++p) bb[p] = Mma<T>::ld(lds_in, pixrow[p] + taprow[tap], kk, khalf);
      };
      load(sbuf, 0, fa[0], fb[0]);
      for (int g = 0; g < ng; ++g) {
        const char* lds_in = sbuf + g * t.buf_bytes;
#pragma unroll
        for (int step = 0; step < NS; ++step) {
          const int cur = step & 1, nxt = cur ^ 1;
          if (step + 1 < NS) load(lds_in, step + 1, fa[nxt], fb[nxt]);
          else if (g + 1 < ng) load(lds_in + t.buf_bytes, 0, fa[nxt], fb[nxt]);
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int p = 0; p < PT; ++p) Mma<T>::mma(fa[cur][c], fb[cur][p], acc[c][p]);
          if (more && (st + 1) * G + g < nloc) {
#pragma unroll
            for (int u = 0; u < SPP; ++u) issue_slot(cbeg + (st + 1) * G + g, nbuf + g * t.buf_bytes, step * SPP + u);
          }
          // keep the prefetch where it is: without the fence the scheduler sinks the reads next to their MFMA
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NS & 1) {   // odd step count: the prefetched fragments of the next chunk sit in slot 1
#pragma unroll
          for (int c = 0; c < CT; ++c) fa[0][c] = fa[1][c];
#pragma unroll
          for (int p = 0; p < PT; ++p) fb[0][p] = fb[1][p];
        }
      }
      }   // (not f32x3)
    }
    __syncthreads();
    HRP_CSTAMP(4);

    // ---- epilogue: accumulators -> LDS tile [pixel][cout] (element type T) ----------------------
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int cl = wc * 32 * CT + c * 32 + 8 * q4 + 4 * khalf;  // first of 4 consecutive local couts
        float bia[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (d.bias || d.scale) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int co = co0 + cl + i;
            if (co < d.Cout) {
              if (d.bias && kz == 0) bia[i] = d.bias[co];
              if (d.scale) { sc[i] = d.scale[co]; sh[i] = d.shift[co]; }
            }
          }
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          const int m = wp * (32 * PT) + p * 32 + l31;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[c][p][4 * q4 + i];
          if (d.bias || d.scale) {   // uniform: the plain (train-mode) conv skips 2 VALU per value
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (v[i] + bia[i]) * sc[i] + sh[i];
          }
          char* dst = lds_out + m * OP + cl * SZ;
          if constexpr (SZ == 4) {
            *(uint4*)dst = Elem<float>::pack(v);
          } else {
            uint2 r;
            r.x = pack_bf2(v[0], v[1]);
            r.y = pack_bf2(v[2], v[3]);
            *(uint2*)dst = r;
          }
        }
      }
    }
    __syncthreads();
    HRP_CSTAMP(5);

    // ---- coalesced pass: residual, ReLU, statistics, global store -------------------------------
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
    const int co = co0 + cv * VEC;
    char* yg = (char*)d.y;
    const char* rg = (const char*)d.res;
    const long long ybase = ((long long)n0 * d.y_H + (oy0 * d.out_stride + d.out_off_y)) * d.y_W +
                            (ox0 * d.out_stride + d.out_off_x);
    // BatchNorm-backward reduce folded into a data-gradient launch (hrp_conv_desc.bnb_*): mean / invstd of the
    // thread's channels, and every row's BatchNorm input + mask byte requested up front (issued between the stores
    // they would wait for the store queue one by one: +4 us per launch)
    // (not in the persistent instance: its plans live across tiles and 40 more registers would halve its occupancy -
    // the launchers never pick it for such a problem)
    const char* bxg = PERSIST ? nullptr : (const char*)d.bnb_x;
    float bmean[VEC], binv[VEC];
    uint4 bx_pre[KST];
    unsigned bits_pre[KST];
    if (bxg && co < d.Cout) {
#pragma unroll
      for (int q = 0; q < VEC; q += 4) {
        const float4 m4 = *(const float4*)(d.bnb_consts + co + q), i4 = *(const float4*)(d.bnb_consts + d.Cout + co + q);
        bmean[q] = m4.x; bmean[q + 1] = m4.y; bmean[q + 2] = m4.z; bmean[q + 3] = m4.w;
        binv[q] = i4.x; binv[q + 1] = i4.y; binv[q + 2] = i4.z; binv[q + 3] = i4.w;
      }
#pragma unroll
      for (int k = 0; k < KST; ++k) {
        bx_pre[k] = make_uint4(0, 0, 0, 0);
        bits_pre[k] = 0;
        if (out_code[k] & cls) continue;
        const size_t opix = (size_t)(ybase + out_rel[k]);
        bx_pre[k] = *(const uint4*)(bxg + (opix * d.bnb_x_pitch + co) * SZ);
        bits_pre[k] = d.bnb_mask[opix * d.bnb_mask_pitch + co / VEC];
      }
    }
    if (co < d.Cout) {
      const bool full = t.vec_ok && (co + VEC <= d.Cout);
#pragma unroll
      for (int k = 0; k < KST; ++k) {
        if (out_code[k] & cls) continue;
        const int m = tid / NV + k * (256 / NV);
        const size_t opix = (size_t)(ybase + out_rel[k]);
        const uint4 raw = *(const uint4*)(lds_out + m * OP + cv * 16);
        float f[VEC];
        Elem<T>::unpack(raw, f);
        if constexpr (SZ == 4) {
          if (t.ksplit > 1) {
            // split-K partial: fp32 atomics into y (zeroed by the launcher, or holding the value to accumulate
            // onto when res == y); slice 0 also adds a separate residual
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              if (co + i < d.Cout) {
                float val = f[i];
                if (rg && rg != (const char*)yg && kz == 0) val += Elem<T>::ld(rg, opix * d.res_pitch + co + i);
                atomicAdd((float*)yg + opix * d.y_pitch + co + i, val);
              }
            }
            continue;
          }
        }
        if (full && !rg && !d.relu) {
          // plain conv output (the train-mode case): the LDS image is already the stored value
          *(uint4*)(yg + (opix * d.y_pitch + co) * SZ) = raw;
          if (bxg) {
            // sum g, sum g * xhat over the masked gradient (the values as stored, like the separate reduce pass)
            const unsigned bits = bits_pre[k];
            float xf[VEC];
            Elem<T>::unpack(bx_pre[k], xf);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const float g = (bits >> i) & 1u ? f[i] : 0.f;
              s1[i] += g;
              s2[i] += g * (xf[i] - bmean[i]) * binv[i];
            }
          } else if (d.stats) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) { s1[i] += f[i]; s2[i] += f[i] * f[i]; }
          }
        } else if (full) {
          if (rg) {
            float r[VEC];
            Elem<T>::unpack(*(const uint4*)(rg + (opix * d.res_pitch + co) * SZ), r);
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] += r[i];
          }
          if (d.relu) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] = fmaxf(f[i], 0.f);
          }
          uint4 packed = Elem<T>::pack(f);
          *(uint4*)(yg + (opix * d.y_pitch + co) * SZ) = packed;
          if (d.stats) {
            float g[VEC];
            Elem<T>::unpack(packed, g);  // statistics of the values as stored
#pragma unroll
            for (int i = 0; i < VEC; ++i) { s1[i] += g[i]; s2[i] += g[i] * g[i]; }
          }
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            if (co + i < d.Cout) {
              float val = f[i];
              if (rg) val += Elem<T>::ld(rg, opix * d.res_pitch + co + i);
              if (d.relu) val = fmaxf(val, 0.f);
              Elem<T>::st(yg, opix * d.y_pitch + co + i, val);
              if (d.stats) {
                float g = Elem<T>::ld(yg, opix * d.y_pitch + co + i);
                s1[i] += g; s2[i] += g * g;
              }
            }
          }
        }
      }
    }
    HRP_CSTAMP(6);
    if (d.stats) {
      // lanes cv, cv + NV, ... of a wave own the same channels: inside a row of 16 lanes they are folded
      // with DPP rotations (VALU speed), across rows with two shuffles; then one partial per wave in LDS
      // ([wave][2][BN]) and one global atomic per channel per workgroup.  (LDS float atomics instead of the
      // per-wave partials were measured 2.5 us slower per tile.)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if constexpr (NV <= 8) { s1[i] += dpp_row_ror<8>(s1[i]); s2[i] += dpp_row_ror<8>(s2[i]); }
        if constexpr (NV <= 4) { s1[i] += dpp_row_ror<4>(s1[i]); s2[i] += dpp_row_ror<4>(s2[i]); }
#pragma unroll
        for (int o = 32; o >= (NV > 16 ? NV : 16); o >>= 1) {
          s1[i] += __shfl_xor(s1[i], o, 64);
          s2[i] += __shfl_xor(s2[i], o, 64);
        }
      }
      if (lane < NV) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          lds_stats[(wave * 2 + 0) * BN + lane * VEC + i] = s1[i];
          lds_stats[(wave * 2 + 1) * BN + lane * VEC + i] = s2[i];
        }
      }
      __syncthreads();
      for (int i = tid; i < 2 * BN; i += 256) {
        const int which = i / BN, ch = i - which * BN;
        if (co0 + ch < d.Cout) {
          float v = lds_stats[(0 * 2 + which) * BN + ch] + lds_stats[(1 * 2 + which) * BN + ch] +
                    lds_stats[(2 * 2 + which) * BN + ch] + lds_stats[(3 * 2 + which) * BN + ch];
          double* slot = d.stats + stat_slot * 2 * d.Cout;
          atomicAdd(&slot[which * d.Cout + co0 + ch], (double)v);
        }
      }
    }
    if constexpr (PERSIST) __syncthreads();   // the next tile's DMA overwrites the output image / statistics partials
    HRP_CSTAMP(7);
  }
}

template <typename T, int CT, int PT, int WC, int WP, int NT, bool PERSIST, bool FAST>
__global__ __launch_bounds__(256) void conv_tile_kernel(const hrp_conv_desc d, const ConvTiling t) {
  conv_tile_body<T, CT, PT, WC, WP, NT, PERSIST, FAST>(d, t, blockIdx.x, gridDim.x, blockIdx.x & (HRP_STAT_SLOTS - 1));
}

// ---------------------------------------------------------------------------------------------
static thread_local int g_conv_lds_budget_kb = 76;
static thread_local bool g_conv_border_reject = false;   // plan_cfg: a configuration was turned down because its tiles are smaller than the tap halo

// Host side: tile geometry, LDS layout and block count of one problem for one tile configuration (no HIP call).
// -> HRP_OK, -100 (this configuration does not fit: try the next) or HRP_ERR_ARG.
template <typename T, int CT, int PT, int WC, int WP, int NT>
static int plan_cfg(const hrp_conv_desc& d, ConvTiling& t, int& lds_out, bool allow_ksplit) {
  constexpr int BN = 32 * CT * WC, BM = 32 * PT * WP;
  constexpr int SZ = Elem<T>::SZ;
  t = ConvTiling{};
  // tile geometry: TW x TH x TI <= BM
  int TW = 1; while (TW < d.Wo && TW < 16) TW <<= 1;
  int TH = 1; while (TH < d.Ho && TH * TW < BM) TH <<= 1;
  int TI = BM / (TW * TH);
  if (TI > d.N) TI = d.N;
  t.TW = TW; t.TH = TH;
  int mindy = 1 << 30, maxdy = -(1 << 30), mindx = 1 << 30, maxdx = -(1 << 30);
  for (int i = 0; i < d.ntaps; ++i) {
    mindy = d.dy[i] < mindy ? d.dy[i] : mindy; maxdy = d.dy[i] > maxdy ? d.dy[i] : maxdy;
    mindx = d.dx[i] < mindx ? d.dx[i] : mindx; maxdx = d.dx[i] > maxdx ? d.dx[i] : maxdx;
  }
  t.mindy = mindy; t.mindx = mindx;
  t.IHt = (TH - 1) * d.in_stride + (maxdy - mindy) + 1;
  t.IWt = (TW - 1) * d.in_stride + (maxdx - mindx) + 1;
  const int budget = g_conv_lds_budget_kb * 1024;  // 76: two workgroups per CU; a lower first try leaves room for a third
  t.w_pieces = NT * BN / 32;
  {  // shrink the number of images per tile until two stage buffers fit
    int per_img = t.IHt * t.IWt * ROW;
    int maxti = (budget / 2 - t.w_pieces * 1024 - 1024) / per_img;
    if (maxti < 1) return -100;
    if (TI > maxti) TI = maxti;
  }
  t.TI = TI;
  t.in_rows = TI * t.IHt * t.IWt;
  t.in_pieces = cdiv(t.in_rows * ROW, 1024);
  if (t.in_pieces > 40) return -100;  // MAXP_IN pieces per wave
  t.buf_bytes = (t.in_pieces + t.w_pieces) * 1024;
  t.tiles_x = cdiv(d.Wo, TW); t.tiles_y = cdiv(d.Ho, TH); t.tiles_n = cdiv(d.N, TI);
  t.n_cout_blk = cdiv(d.Cout, BN);
  t.nblocks = t.tiles_x * t.tiles_y * t.tiles_n * t.n_cout_blk;
  // split-K for skinny fp32 layers (the fully connected heads: 64 rows x 1024..2056 inputs give 16 workgroups
  // otherwise): K slices in separate workgroups, partial sums leave as fp32 atomics
  t.ksplit = 1;
  {
    const int nch = cdiv(d.Cin * SZ, ROW);
    t.cps = nch;
    // (out_stride == 1: the launch covers all of y, which it zeroes first)
    // (not onto an existing value - res == y, a gradient that accumulates: (y + a) + b and (y + b) + a differ in the last bit)
    if (allow_ksplit && SZ == 4 && !d.relu && !d.scale && !d.stats && d.out_stride == 1 && d.y_H == d.Ho && d.y_W == d.Wo && t.nblocks <= 64 &&
        (const void*)d.res != (const void*)d.y &&
        // y must own its whole pitch: the zero fill below runs over N * y_H * y_W * y_pitch elements from d.y - a channel SLICE of a
        // wider buffer (ASPP's branches into the 1280-channel concatenation, PlanBuilder.channel_slice) would have its neighbours'
        // columns wiped and the fill run past the end of the allocation
        d.y_pitch <= ((d.Cout + 7) / 8) * 8) {
      int ks = 1;
      // at most TWO slices: two fp32 partials added atomically onto a zeroed output give the same bits in either order
      // (a + b == b + a); four or more do not - the fp32 inference forward differed in the last bit from run to run
      // (256-channel fuse layers at 8 x 8), which moved key-point 0 by up to 1.5e-3 px.  The 1 x 1 layers on pooled
      // features, which wanted 16 slices, run on the linear kernels with their ordered reduction since round 3.
      static const int ks_max = 2;
      while (ks < ks_max && t.nblocks * ks * 2 <= 256 && nch / (ks * 2) >= 8) ks *= 2;
      if (ks > 1) { t.ksplit = ks; t.cps = cdiv(nch, ks); t.nblocks *= ks; }
    }
  }
  // the DMA / store plans classify validity per tile row / column class: only the first and the last tile
  // row (column) may reach outside the image
  if ((t.tiles_y >= 2 && (TH * d.in_stride + mindy < 0 ||
                          (t.tiles_y - 2) * TH * d.in_stride + mindy + t.IHt - 1 >= d.H)) ||
      (t.tiles_x >= 2 && (TW * d.in_stride + mindx < 0 ||
                          (t.tiles_x - 2) * TW * d.in_stride + mindx + t.IWt - 1 >= d.W))) {
    // (a tile smaller than the halo - dilated taps: the next, larger configuration may do)
    g_conv_border_reject = true;
    return -100;
  }
  const int out_bytes = BM * (BN * SZ + 16);
  {  // chunks per stage: as many as two stages fit in half of LDS
    const int nsub = t.cps;
    // (always half of LDS: a launch with one workgroup per CU shares the CU with kernels of other lanes -
    // taking all of LDS for deeper stages cost 2 ms per step)
    int G = budget / (2 * t.buf_bytes);
    if (G < 1) G = 1;
    if (G > 16) G = 16;
    if (G > nsub) G = nsub;
    if (std::is_same<T, f32x3_t>::value && G >= 2) G &= ~1;      // (chunk pairs share an MFMA: an even stage keeps every pair whole)
    t.G = G;
  }
  int main_bytes = 2 * t.G * t.buf_bytes;
  if (out_bytes > main_bytes) main_bytes = out_bytes;
  t.lds_stats_off = round_up(main_bytes, 16);
  int lds = t.lds_stats_off + 8 * BN * 4;  // [wave][sum, sumsq][BN] partials
  if (lds > 160 * 1024) return -100;
  const bool aligned = ((uintptr_t)d.y % 16 == 0) && ((size_t)d.y_pitch * SZ % 16 == 0) &&
                       (!d.res || (((uintptr_t)d.res % 16 == 0) && ((size_t)d.res_pitch * SZ % 16 == 0)));
  t.vec_ok = aligned ? 1 : 0;
  t.fd_ihw = make_fastdiv(t.IHt * t.IWt); t.fd_iwt = make_fastdiv(t.IWt);
  t.fd_thw = make_fastdiv(TH * TW); t.fd_tw = make_fastdiv(TW);
  t.fd_ncb = make_fastdiv(t.n_cout_blk); t.fd_tx = make_fastdiv(t.tiles_x); t.fd_ty = make_fastdiv(t.tiles_y);
  lds_out = lds;
  return HRP_OK;
}

template <typename T, int CT, int PT, int WC, int WP, int NT>
static int launch_cfg(const hrp_conv_desc& d, hipStream_t s) {
  constexpr int SZ = Elem<T>::SZ;
  ConvTiling t;
  int lds = 0;
  const int prc = plan_cfg<T, CT, PT, WC, WP, NT>(d, t, lds, true);
  if (prc != HRP_OK) return prc;
  // persistent workgroups (about two per CU, each walking tiles b, b + grid, ...) when a tile is little MFMA
  // work and there are several tiles per CU; one tile per workgroup otherwise
  const int mfma_per_tile = cdiv(d.Cin * SZ, ROW) * NT * CT * PT * Mma<T>::KSTEPS;
  // (off by default since the end of round 1: the persistent instances hold 196-208 registers per lane against 96-132,
  // so nothing of another lane fits next to their two workgroups on a CU; alone they are 15 % faster on 32->32 @64x64,
  // in the step they cost 0.55 ms - HRP_CONV_PERSIST=48 restores them)
  static const int persist_max = 0;
  constexpr bool CAN_PERSIST = NT == 9;   // (only 3x3 layers have tiles small enough to profit; keeps the instantiation count down)
  const bool persist = CAN_PERSIST && mfma_per_tile <= persist_max && t.nblocks >= 3 * 256 && !d.bnb_x;   // 1x1 layers are store bound: many small workgroups
  const bool fastp = (d.Cin * SZ) % ROW == 0;   // no half-filled last chunk: per-piece advancing pointers
  void (*kern)(const hrp_conv_desc, const ConvTiling) =
      fastp ? conv_tile_kernel<T, CT, PT, WC, WP, NT, false, true> : conv_tile_kernel<T, CT, PT, WC, WP, NT, false, false>;
  if constexpr (CAN_PERSIST) {
    if (persist) kern = fastp ? conv_tile_kernel<T, CT, PT, WC, WP, NT, true, true> : conv_tile_kernel<T, CT, PT, WC, WP, NT, true, false>;
  }
  static bool attr_set[4] = {false, false, false, false};
  if (!attr_set[persist * 2 + fastp]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[persist * 2 + fastp] = true;
  }
  int occ = (160 * 1024) / lds;
  occ = occ < 1 ? 1 : occ > 2 ? 2 : occ;
  static const int pgrid = 0;   // persistent workgroups per CU (0: off; swept: DESIGN 5)
  const int pocc = pgrid > 0 ? pgrid : occ;
  const int grid = persist && t.nblocks > 256 * pocc ? 256 * pocc : t.nblocks;
  if (t.ksplit > 1 && d.res != d.y)
    zero_async(d.y, (size_t)d.N * d.y_H * d.y_W * d.y_pitch * SZ, s);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, d, t);
  return check_launch("conv_tile_kernel");
}

// Tile choice.  Weight bytes streamed per workgroup are fixed by BN x K, so the pixel tile should be as
// large as the launch allows while still giving every CU a workgroup (>= 256 blocks); BN = 64 at most so
// that two stage buffers of a 3x3 conv stay under ~60 KB.
template <typename T, int NT>
static int launch_conv_nt(const hrp_conv_desc& d, hipStream_t s) {
  const long pixels = (long)d.N * d.Ho * d.Wo;
  static const long want = 256;
  static const int low_kb = 76;
  int rc = -100;
  g_conv_border_reject = false;
  for (int pass = 0; pass < 2 && rc == -100; ++pass) {
  g_conv_lds_budget_kb = pass == 0 ? low_kb : 76;
  if (pass == 1 && low_kb >= 76) break;
  if (d.Cout <= 32) {
    if (pixels / 256 >= want) rc = launch_cfg<T, 1, 2, 1, 4, NT>(d, s);
    if (rc == -100) rc = launch_cfg<T, 1, 1, 1, 4, NT>(d, s);
  } else {
    const long nb = (d.Cout + 63) / 64;
    if (pixels / 256 * nb >= want) rc = launch_cfg<T, 2, 2, 1, 4, NT>(d, s);
    if (rc == -100 && pixels / 128 * nb >= want) rc = launch_cfg<T, 2, 1, 1, 4, NT>(d, s);
    if (rc == -100) rc = launch_cfg<T, 1, 1, 2, 2, NT>(d, s);
    if (rc == -100) rc = launch_cfg<T, 2, 1, 1, 4, NT>(d, s);
    if (rc == -100) rc = launch_cfg<T, 1, 1, 1, 4, NT>(d, s);   // 32 couts per workgroup: halves the weight slab (16-tap kernels)
  }
  }
  g_conv_lds_budget_kb = 76;
  if (rc == -100) {
    if (g_conv_border_reject)
      set_error("conv: no tile configuration fits: the tap offsets reach beyond the border tiles of every tile that fits LDS "
                "(H=%d W=%d Ho=%d Wo=%d Cin=%d stride=%d taps=%d)", d.H, d.W, d.Ho, d.Wo, d.Cin, d.in_stride, d.ntaps);
    else
      set_error("conv: tile does not fit LDS (H=%d W=%d Cin=%d stride=%d taps=%d)", d.H, d.W, d.Cin, d.in_stride, d.ntaps);
    return HRP_ERR_ARG;
  }
  return rc;
}

template <typename T>
static int launch_conv(const hrp_conv_desc& d, hipStream_t s) {
  switch (d.ntaps) {
    case 1: return launch_conv_nt<T, 1>(d, s);
    case 2: return launch_conv_nt<T, 2>(d, s);
    case 4: return launch_conv_nt<T, 4>(d, s);
    case 9: return launch_conv_nt<T, 9>(d, s);
    case 16: return launch_conv_nt<T, 16>(d, s);   // 4x4 kernels: ResNet stem after space-to-depth, ConvTranspose2d backward
    default:
      set_error("conv: ntaps=%d is not one of the built tap counts (1, 2, 4, 9, 16)", d.ntaps);
      return HRP_ERR_ARG;
  }
}

}  // namespace hrp

