// Row-RING 3x3 convolution for the high-resolution BasicBlock layers (C = 32 @ W = 64, C = 64 @ W = 32; reference
// HRnet.py:28-57): the software-pipelined successor of conv_row_body (conv_row.h), same descriptor, same fused BatchNorm
// forms, same LDS row layout / bank swizzle / MFMA row -> channel permutation / epilogue helpers.
//
// What changes is the SCHEDULE (the strip kernel ran load -> prologue -> MFMA -> epilogue one after the
// other inside a workgroup that lived for one or two strips; its waves sat in s_waitcnt half of their cycles):
//   * a workgroup is PERSISTENT over a band of rows of one image (16 .. 64 rows) and walks it top to bottom in STEPS of 4
//     output rows; the input rows live in a ring of RX = 12 LDS row slots - every input row is fetched ONCE per band (the
//     strip kernel re-fetched 2 of every 10 rows), weights and per-channel constants are set up once per band;
//   * the direct-to-LDS DMA of step i + 1 (4 rows) and the register loads of every fused operand of step i + 1 / of the
//     epilogue of step i are ISSUED BEFORE the MFMA loop of step i and land underneath it; after the loop the wave
//     transforms the newly landed rows in place (BatchNorm prologues), then runs the epilogue of step i (its stores get a
//     whole step to retire), then ONE barrier;
//   * per step a wave owns 2 output rows x 32 pixels x 32 output channels (conv_block.h's role-0 tiling): 18 KS MFMAs from
//     12 KS LDS reads; LDS footprint 52 KiB (the strip kernel: 46 - 62 KiB), so two workgroups - or one and a workgroup of
//     the other lane's kernel - share a CU.
// Statistics: C = 32 keeps its per-lane partial sums in registers over the whole band and runs ONE reduce-scatter butterfly per
// workgroup (the strip kernel: one per strip); C = 64 (144 weight registers) reduces per step.
#pragma once

namespace hrp {

template <int C>
struct RingCfg {
  using R0 = RowCfg<C>;
  static constexpr int W = R0::W, P = R0::P, S = R0::S, KS = R0::KS, MT = R0::MT, NCOL = R0::NCOL, PXP = R0::PXP, ROWB = R0::ROWB;
  static constexpr int RX = 12;                                     // ring rows: [4 i, 4 i + 6) read by step i, [4 i + 6, 4 i + 10) in flight
  static constexpr int TILE_BYTES = P + RX * ROWB;                  // leading zero pixel + ring rows (each followed by one zero pixel)
  static constexpr int CTAB_OFF = (TILE_BYTES + 255) & ~255;        // per-channel constants [10][C] floats (conv_row_body's table)
  static constexpr int STAT_OFF = CTAB_OFF + 10 * C * 4;            // [4 waves][64] floats
  static constexpr int LDS_BYTES = STAT_OFF + 4 * 64 * 4;
  static constexpr bool ACC_STATS = C == 32;                        // per-lane statistic sums stay in registers over the band
};

// Operands of an epilogue form, requested before the MFMA loop (ring_epi_load) and consumed after it (ring_epi_math).  The
// form is a set of WORKGROUP-UNIFORM descriptor tests; both functions evaluate them the same way.  Lean instantiation (!EXT): an
// epilogue reduce has no residual (row_ext), so ONE operand array serves either the reduce's BatchNorm input or the residual.
template <int NT, bool EXT>
struct RingEpi {
  uint4 a[NT][2];                       // !EXT: bnb_x or res; EXT: bnb_x
  uint4 b[EXT ? NT : 1][2];             // EXT: res
  int mb[EXT ? NT : 1], rb[EXT ? NT : 1];
};

template <int NT, bool EXT>
__device__ __forceinline__ void ring_epi_load(const hrp_conv_desc& d, const unsigned (&off)[NT], const bool bnb, RingEpi<NT, EXT>& e) {
  const char* bx = (const char*)d.bnb_x;
  const char* rq = (const char*)d.res;
  if constexpr (!EXT) {
    const char* q = bnb ? bx : rq;
    if (q) {
#pragma unroll
      for (int t = 0; t < NT; ++t) { e.a[t][0] = *(const uint4*)(q + off[t]); e.a[t][1] = *(const uint4*)(q + off[t] + 16); }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      e.mb[t] = 0;
      e.rb[t] = 0xffff;
      if (bnb) {
        e.a[t][0] = *(const uint4*)(bx + off[t]); e.a[t][1] = *(const uint4*)(bx + off[t] + 16);
        if (d.bnb_mask) e.mb[t] = *(const unsigned short*)(d.bnb_mask + (off[t] >> 4));
      }
      if (rq) {
        e.b[t][0] = *(const uint4*)(rq + off[t]); e.b[t][1] = *(const uint4*)(rq + off[t] + 16);
        if (d.res_mask) e.rb[t] = *(const unsigned short*)(d.res_mask + (off[t] >> 4));
      }
    }
  }
}

// The forms of row_epilogue (conv_row.h) on pre-loaded operands.  Arithmetic, rounding points and summation order per lane are
// row_epilogue's (row_epi_bnb_math is called on the same operand struct).
template <int NT, bool EXT>
__device__ __forceinline__ void ring_epi_math(const hrp_conv_desc& d, const f32x16 (&acc)[NT], const unsigned (&off)[NT], const int cl,
                                              const float* ctab, const int C, const bool bnb, const RingEpi<NT, EXT>& e,
                                              float (&s1)[16], float (&s2)[16]) {
  const unsigned ok = (1u << NT) - 1;
  if (bnb) {
    RowEpiOps<NT> o;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      o.xr[t][0] = e.a[t][0]; o.xr[t][1] = e.a[t][1];
      if constexpr (EXT) { o.rr[t][0] = e.b[t][0]; o.rr[t][1] = e.b[t][1]; o.mb[t] = e.mb[t]; o.rb[t] = e.rb[t]; }
    }
    if constexpr (!EXT) row_epi_bnb_math<NT, false, false>(d, acc, off, ok, cl, ctab, C, o, s1, s2);
    else {
      const bool ubits = d.bnb_mask != nullptr, ures = d.res != nullptr;
      if (ubits) { if (ures) row_epi_bnb_math<NT, true, true>(d, acc, off, ok, cl, ctab, C, o, s1, s2); else row_epi_bnb_math<NT, true, false>(d, acc, off, ok, cl, ctab, C, o, s1, s2); }
      else { if (ures) row_epi_bnb_math<NT, false, true>(d, acc, off, ok, cl, ctab, C, o, s1, s2); else row_epi_bnb_math<NT, false, false>(d, acc, off, ok, cl, ctab, C, o, s1, s2); }
    }
    return;
  }
  char* yg = (char*)d.y;
  const bool has_res = d.res != nullptr, aff = d.scale != nullptr, relu = d.relu != 0;
  float sc[16], sh[16];
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
      if (has_res) {
        float r[8];
        if constexpr (EXT) {       // (rb = 0xffff without a residual mask)
          Elem<bf16_t>::unpack(e.b[t][hh], r);
#pragma unroll
          for (int i = 0; i < 8; ++i) r[i] = row_keep_if_bit(r[i], e.rb[t], 8 * hh + i);
        } else {
          Elem<bf16_t>::unpack(e.a[t][hh], r);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += r[i];
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
      }
      const uint4 pk = Elem<bf16_t>::pack(v);
      *(uint4*)(yg + off[t] + 16 * hh) = pk;
      Elem<bf16_t>::unpack(pk, v);            // statistics of the values as stored (unconditional: a branch here makes s1 / s2
#pragma unroll                               //  loop-carried through two paths and costs 32 register copies per step)
      for (int i = 0; i < 8; ++i) { s1[8 * hh + i] += v[i]; s2[8 * hh + i] = fmaf(v[i], v[i], s2[8 * hh + i]); }
    }
  }
}

// PRO: the descriptor's pro_mode as a compile-time constant (the dispatcher below branches on it once per workgroup): a launch
// without a prologue carries no second-operand registers through its MFMA loop.
template <int C, int PRO, bool EXT>
__device__ __forceinline__ void conv_ring_body_t(const hrp_conv_desc& d, const RowPlan& rp, int bid, const int stat_slot) {
  using R = RingCfg<C>;
  constexpr int W = R::W, P = R::P, S = R::S, KS = R::KS, ROWB = R::ROWB, RX = R::RX;
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

  // ---- the band of this workgroup: rp.nstrips = N * rp.spi bands of rp.spw rows; the bands of one image on one XCD (their seam
  // rows are fetched twice: the second fetch finds them in that XCD's L2)
  if ((rp.nstrips & 7) == 0) bid = (bid & 7) * (rp.nstrips >> 3) + (bid >> 3);
  const int n = fdiv(bid, rp.fd_spi), band = bid - n * rp.spi;
  const int H = d.H, brows = rp.spw, r0 = band * brows, nB = brows >> 2;
  const unsigned img_off = (unsigned)n * (unsigned)(H * W * P);
  const bool ub = EXT && d.pro_mask != nullptr, wgm = EXT && d.pro_side2 != nullptr;     // (uniform: RowPro::bwd)
  const bool bnb = d.bnb_x != nullptr;

  // ---- staging: piece `wave` of every row; the lane's 16 bytes = (pixel lane / S of the piece, slot lane % S), holding the
  // LOGICAL slot (lane % S) ^ g(x).  Ring index q <-> image row r0 - 1 + q, ring slot q mod RX (tracked incrementally).
  const int px_in_piece = lane / S;
  const int xcol = wave * R::PXP + px_in_piece;
  const int lslot = (lane % S) ^ R::R0::g(xcol);
  const unsigned lane_off = (unsigned)(wave * 1024 + px_in_piece * P + lslot * 16);
  char* ring = smem + P;
  auto wrap = [](const int v) { return v >= RX ? v - RX : v; };
  // CNT rows from ring index qi into slots slot, slot + 1, .. (mod RX); pro_mode 2 also requests the second operand (the
  // BatchNorm input, and the mask byte) of the same lane position into registers
  auto stage = [&](const int qi, int slot, auto cnt_c, uint4* x2v, int* bitv) {
    constexpr int CNT = decltype(cnt_c)::value;
    const char* xg = (const char*)d.x + img_off + lane_off;
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
      const int r = r0 - 1 + qi + k;
      char* dst = ring + slot * ROWB + wave * 1024;
      if (r >= 0 && r < H) dma16(xg + r * (W * P), dst);
      else *(uint4*)(dst + lane * 16) = make_uint4(0, 0, 0, 0);
      slot = wrap(slot + 1);
    }
    if constexpr (PRO == 2) {
      const char* x2g = (const char*)d.pro_x2 + img_off + lane_off;
#pragma unroll
      for (int k = 0; k < CNT; ++k) {
        const int r = r0 - 1 + qi + k;
        x2v[k] = make_uint4(0, 0, 0, 0);
        if constexpr (EXT) bitv[k] = -1;
        if (r >= 0 && r < H) {
          x2v[k] = *(const uint4*)(x2g + r * (W * P));
          if constexpr (EXT) {
            if (d.pro_mask) bitv[k] = d.pro_mask[((img_off + lane_off) >> 4) + r * (W * P / 16)];
          }
        }
      }
    }
  };
  // the BatchNorm prologue of the rows just staged, in place: every lane transforms exactly the 16 bytes it DMA'd itself (no
  // barrier between landing and transform); side outputs only for the band's OWN rows (a seam row belongs to the neighbour)
  auto transform = [&](const int qi, int slot, auto cnt_c, const uint4* x2v, const int* bitv) {
    constexpr int CNT = decltype(cnt_c)::value;
    RowPro pc;
    int cb = lslot * 8;                                         // the lane's 8 channels
    asm volatile("" : "+v"(cb));                                // (opaque per step: the constants are re-read from LDS, not kept
    pc.load(ctab, C, cb);                                       //  in registers across the MFMA loop)
    if constexpr (PRO == 2) pc.load2(ctab, C, cb);
    char* side = (char*)d.pro_side;
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
      const int r = r0 - 1 + qi + k;
      char* p = ring + slot * ROWB + wave * 1024 + lane * 16;
      slot = wrap(slot + 1);
      if (r < 0 || r >= H) continue;
      const bool own = r >= r0 && r < r0 + brows;
      const unsigned off = img_off + lane_off + r * (W * P);
      if constexpr (PRO == 1) {
        const uint4 o = pc.act(*(const uint4*)p);
        *(uint4*)p = o;
        if (side && own) *(uint4*)(side + off) = o;
      } else {
        uint4 gm;
        const uint4 o = pc.template bwd<EXT>(*(const uint4*)p, x2v[k], EXT ? bitv[k] : -1, gm, ub, wgm);
        *(uint4*)p = o;
        if (own) {
          if (side) *(uint4*)(side + off) = o;
          if constexpr (EXT) {
            if (d.pro_side2) row_side2(d, off, gm);
          }
        }
      }
    }
  };
  {
    uint4 x2i[PRO == 2 ? 6 : 1];
    int biti[PRO == 2 && EXT ? 6 : 1];
    stage(0, 0, std::integral_constant<int, 6>{}, x2i, biti);
    // the zero pixels: one in front of ring slot 0, one behind every ring slot
    if (tid < (RX + 1) * S) {
      const int k = tid / S, j = tid - k * S;
      *(uint4*)(smem + (k == 0 ? 0 : P + (k - 1) * ROWB + W * P) + j * 16) = make_uint4(0, 0, 0, 0);
    }
    // ---- per-channel constants (LDS table [10][C], conv_row_body's rows)
    if (PRO != 0 && tid < C) {
      float mean, inv, sc, sh;
      row_bn_consts(d.pro_stats, d.pro_gamma, d.pro_beta, d.pro_count, d.pro_eps, tid, C, mean, inv, sc, sh);
      ctab[0 * C + tid] = sc; ctab[1 * C + tid] = sh;
      if (PRO == 2) {
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the six rows of step 0 have landed
    if constexpr (PRO != 0) {
      __syncthreads();                                            // the constant table
      transform(0, 0, std::integral_constant<int, 6>{}, x2i, biti);
    }
  }

  // ---- weights: A fragments of this wave's 32 output channels, resident for the band (conv_row_body: MFMA row rho = 8 q + 4 h + i
  // carries output channel 16 h + 4 q + i, so that a lane's 16 accumulators are 16 consecutive channels of one pixel).  Requested
  // after the first rows' prologue, whose registers they would otherwise share.
  bf16x8 wf[9][KS];
  {
    const int co_lane = m * 32 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    const char* wl = (const char*)d.w + co_lane * ROW + half * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
        wf[t][kk] = *(const bf16x8*)(wl + (size_t)((kk * d.w_ntaps + rp.wslot[t]) * C) * ROW);
    // (a use the compiler can see: its wait for these loads belongs HERE - left to the first MFMAs it becomes a vmcnt(small) inside
    // the loop body, which in every later step drains the row DMA and the epilogue operands the loop is supposed to run over)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) asm volatile("" :: "v"(wf[t][kk]));
  }
  __syncthreads();
  HRP_CSTAMP(2);

  // read address of (dx, kk = 0) in ring slot 0: pixel x = col*32 + l31 + dx (x = -1 / W are the shared zero pixels); the K chunk
  // kk is an XOR of bits 5.. (slot' = (2 kk + half) ^ g(x))
  int a0[3];
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi) {
    const int xq = col * 32 + l31 + dxi - 1;
    a0[dxi] = P + xq * P + ((half ^ R::R0::g(xq)) << 4);
  }
  const int cl = m * 32 + 16 * half;      // first output channel of the lane
  const unsigned out0 = img_off + (unsigned)((r0 + 2 * rg) * W + col * 32 + l31) * P + cl * 2;
  float s1[16], s2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
  float vtot = 0.f;                       // (C = 64: statistics of the steps so far, already reduced over the half wave)

  int xb = 0;                             // ring slot of ring index 4 i
  for (int i = 0; i < nB; ++i) {
    const bool more = i + 1 < nB;
    // ---- requests that land under the MFMA loop: the operands of this step's epilogue, then the rows of the next step
    unsigned off[2];
    off[0] = out0 + (unsigned)(4 * i) * (W * P);
    off[1] = off[0] + W * P;
    RingEpi<2, EXT> eo;
    ring_epi_load<2, EXT>(d, off, bnb, eo);
    uint4 x2v[PRO == 2 ? 4 : 1];
    int bitv[PRO == 2 && EXT ? 4 : 1];
    if (more) stage(4 * i + 6, wrap(xb + 6), std::integral_constant<int, 4>{}, x2v, bitv);

    // ---- MFMA loop: input rows ring index 4 i + 2 rg .. + 3 -> output rows 4 i + 2 rg, + 1 of the band
    f32x16 acc[2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
    {
      int roff[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) roff[k] = wrap(wrap(xb + 2 * rg) + k) * ROWB;
      constexpr int NSTEP = 4 * 3 * KS, RINGB = 4, AHEAD = 3;
      bf16x8 bq[RINGB];
      auto rd = [&](int s) -> bf16x8 {   // s is a constant after unrolling
        const int irel = s / (3 * KS), dxi = (s / KS) % 3, kk = s % KS;
        return *(const bf16x8*)(smem + roff[irel] + (a0[dxi] ^ (kk << 5)));
      };
#pragma unroll
      for (int s = 0; s < AHEAD; ++s) bq[s % RINGB] = rd(s);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        if (s + AHEAD < NSTEP) bq[(s + AHEAD) % RINGB] = rd(s + AHEAD);
        const int irel = s / (3 * KS), dxi = (s / KS) % 3, kk = s % KS;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int dyi = irel - o;               // input row irel = output row o + dy, dy = dyi - 1
          if (dyi >= 0 && dyi <= 2)
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[dyi * 3 + dxi][kk], bq[s % RINGB], acc[o], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);        // keep the read-ahead where it is
      }
    }
    if (i == 0) HRP_CSTAMP(4);

    // ---- everything requested above has landed; the next step's rows get their prologue (other waves may still be in their
    // MFMA loop: they read ring indices < 4 i + 6, this touches >= 4 i + 6)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (PRO != 0) {
      if (more) transform(4 * i + 6, wrap(xb + 6), std::integral_constant<int, 4>{}, x2v, bitv);
    }

    // ---- epilogue of this step (stores retire under the next step)
    ring_epi_math<2, EXT>(d, acc, off, cl, ctab, C, bnb, eo, s1, s2);
    if constexpr (!R::ACC_STATS) {
      if (d.stats) {
        vtot += row_reduce32(s1, s2, l31);
#pragma unroll
        for (int e = 0; e < 16; ++e) s1[e] = s2[e] = 0.f;
      }
    }
    if (i == 0) HRP_CSTAMP(5);
    __syncthreads();
    xb = wrap(xb + 4);
  }
  if (d.stats) {
    if constexpr (R::ACC_STATS) vtot = row_reduce32(s1, s2, l31);
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
__device__ __forceinline__ void conv_ring_body(const hrp_conv_desc& d, const RowPlan& rp, const int bid, const int stat_slot) {
  const int pro = d.pro_mode;
  if (row_ext(d)) {     // (the extended options only occur with the backward prologue, pro_mode 2 - and in hand-made descriptors)
    if (pro == 2) conv_ring_body_t<C, 2, true>(d, rp, bid, stat_slot);
    else if (pro == 1) conv_ring_body_t<C, 1, true>(d, rp, bid, stat_slot);
    else conv_ring_body_t<C, 0, true>(d, rp, bid, stat_slot);
  } else {
    if (pro == 2) conv_ring_body_t<C, 2, false>(d, rp, bid, stat_slot);
    else if (pro == 1) conv_ring_body_t<C, 1, false>(d, rp, bid, stat_slot);
    else conv_ring_body_t<C, 0, false>(d, rp, bid, stat_slot);
  }
}

}  // namespace hrp
