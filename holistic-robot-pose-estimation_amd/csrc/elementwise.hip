// Fused element-wise op of the HRNet graph and its backward (HBM-bound; 16-byte vector loads/stores).
//
//   out[n,y,x,c] = act( sum_j f_j( in_j[n, y/up_j, x/up_j, c] ) )
//
// covers BatchNorm (+ReLU) after a conv, the residual add of BasicBlock / Bottleneck, the whole
// cross-resolution fuse of HighResolutionModule (BN of each incoming path + nearest upsample + sum +
// ReLU in ONE pass; reference HRnet.py:256-263 materialises every term), and the cls-head adds.
// Train-mode BN needs no separate finalize kernel: each workgroup derives scale/shift of its channel slab
// once (into LDS) from the (sum, sumsq) statistic slots the producing conv accumulated in its epilogue.
#include "hrp_common.h"
#include "batch.h"
#include <stdlib.h>
#include <string.h>

namespace hrp {

template <typename T, int V>
struct VecIO {
  static_assert(V == 1 || V == Elem<T>::VEC, "vector width");
  __device__ static __forceinline__ void ld(const void* p, size_t i, float* f) {
    if constexpr (V == 1) f[0] = Elem<T>::ld(p, i);
    else Elem<T>::unpack(*(const uint4*)((const char*)p + i * Elem<T>::SZ), f);
  }
  __device__ static __forceinline__ void st(void* p, size_t i, const float* f) {
    if constexpr (V == 1) Elem<T>::st(p, i, f[0]);
    else *(uint4*)((char*)p + i * Elem<T>::SZ) = Elem<T>::pack(f);
  }
  // load now, convert at the use: a batch of bf16 vectors waits in half the registers of its fp32 form
  struct Raw { uint4 q; };
  __device__ static __forceinline__ Raw ldr(const void* p, size_t i) {
    Raw r;
    if constexpr (V == 1) r.q = make_uint4(__float_as_uint(Elem<T>::ld(p, i)), 0, 0, 0);
    else r.q = *(const uint4*)((const char*)p + i * Elem<T>::SZ);
    return r;
  }
  __device__ static __forceinline__ void cvt(const Raw& r, float* f) {
    if constexpr (V == 1) f[0] = __uint_as_float(r.q.x);
    else Elem<T>::unpack(r.q, f);
  }
};

// x rotated right by N lanes inside each row of 16 lanes (DPP row_ror)
template <int N>
__device__ __forceinline__ float ew_row_ror(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}

constexpr int TAB_CH = 512;  // channels per block whose constants are shared through LDS

// scale / shift / mean / invstd of ONE channel of an input
__device__ __forceinline__ void channel_consts(const hrp_ew_input& in, int c, int C, float& sc, float& sh, float& mean, float& inv) {
  sc = 1.f; sh = 0.f; mean = 0.f; inv = 1.f;
  if (c >= C) return;
  if (in.mode == HRP_EW_AFFINE) {
    sc = in.a[c]; sh = in.b[c];
    // the reduce pass of an affine input accumulates sum g * (x * a + b) = sum g * (x - mean) * inv with these (the eval-mode
    // BatchNorm backward passes (a, b) = (invstd, -mean * invstd) of the running statistics: dgamma; a > 0 there)
    inv = sc; mean = sc != 0.f ? -sh / sc : 0.f;
  } else if (in.mode == HRP_EW_BN_TRAIN) {
    float m = slot_sum(in.stats, c, 2 * C) / in.count;
    float var = fmaxf(slot_sum(in.stats, C + c, 2 * C) / in.count - m * m, 0.f);
    float is = rsqrtf(var + in.eps);
    mean = m; inv = is;
    sc = in.a[c] * is;
    sh = in.b[c] - m * sc;
  }
}

// Fill tab[f][e] (f = 0 scale, 1 shift, 2 mean, 3 invstd) for the block's channels cbase .. cbase+nch-1.
// nch <= TAB_CH: computed once per block; otherwise every thread computes its own V channels.
template <int V>
__device__ __forceinline__ void load_consts(const hrp_ew_input& in, int C, int cbase, int nch, int c, float* tab,
                                            float* sc, float* sh, float* mean, float* inv) {
  // tab = 4 rows of nch floats in dynamic LDS (sized by the launcher: a static worst-case table kept 12-28 KiB
  // of LDS per block away from the conv / wgrad workgroups of other lanes sharing the CU)
  if (nch <= TAB_CH) {
    for (int e = threadIdx.x; e < nch; e += 256)
      channel_consts(in, cbase + e, C, tab[e], tab[nch + e], tab[2 * nch + e], tab[3 * nch + e]);
    __syncthreads();
    const int e0 = c - cbase;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int e = min(e0 + i, nch - 1);
      sc[i] = tab[e]; sh[i] = tab[nch + e]; mean[i] = tab[2 * nch + e]; inv[i] = tab[3 * nch + e];
    }
    __syncthreads();  // the table may be reused for the next input
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) channel_consts(in, c + i, C, sc[i], sh[i], mean[i], inv[i]);
  }
}

// MAXIN: inputs this instance can take (2: the blocks' activations, half the channel-constant registers of the
// 4-input fuse instance - the register count decides how many waves of OTHER lanes' kernels fit next to this one)
// (bx, by) of gx x nslab: the block's position in its problem's grid - blockIdx / gridDim in the single launch, decoded
// from the linear block index in the batched launch
// LEAKY: relu == 2 (nn.LeakyReLU(), slope 0.01) - single launches only, so that the ReLU instances stay as they were
template <typename T, int V, int MAXIN, bool LEAKY = false>
__device__ __forceinline__ void ew_fwd_body(const hrp_ew_desc& d, const int tpr, const int bx, const int by, const int gx) {
  extern __shared__ float ew_lds[];
  float* tab = ew_lds;
  const int cv = (by * tpr + threadIdx.x % tpr);
  const int c = cv * V;
  const int cbase = by * tpr * V, nch = tpr * V;
  const int ppb = 256 / tpr;  // pixels per block-iteration
  constexpr float neg = LEAKY ? 0.01f : 0.f;     // the mask / sign logic is ReLU's
  float sc[MAXIN][V], sh[MAXIN][V];
  {
    float m[V], iv[V];
#pragma unroll
    for (int j = 0; j < MAXIN; ++j)
      if (j < d.nin) {
        if (d.in[j].mode == HRP_EW_IDENTITY) {
#pragma unroll
          for (int i = 0; i < V; ++i) { sc[j][i] = 1.f; sh[j][i] = 0.f; }
        } else {
          load_consts<V>(d.in[j], d.C, cbase, nch, c, tab, sc[j], sh[j], m, iv);
          // mean / invstd of input 0 for the backward pass (hrp_conv_desc.bnb_consts): the first pixel row of the
          // first block of every channel slab publishes them
          if (j == 0 && d.consts_out && bx == 0 && (int)threadIdx.x < tpr && c < d.C && d.in[0].mode == HRP_EW_BN_TRAIN) {
#pragma unroll
            for (int i = 0; i < V; ++i) { d.consts_out[c + i] = m[i]; d.consts_out[d.C + c + i] = iv[i]; }
          }
        }
      }
  }
  if (c >= d.C) return;
  // index arithmetic in 32 bits (the launcher checks N*H*W < 2^31); the pixel decomposition - two integer
  // divisions - only when some input is upsampled (64-bit divisions per vector made this kernel VALU bound)
  const unsigned npix = (unsigned)d.N * d.H * d.W;
  bool any_up = false;
#pragma unroll
  for (int j = 0; j < MAXIN; ++j) any_up = any_up || (j < d.nin && d.in[j].up != 1);
  const unsigned uW = d.W, uH = d.H;
  const unsigned stride = gx * ppb;
  unsigned p = bx * ppb + threadIdx.x / tpr;
  if (!any_up && d.nin <= 2) {
    // one or two same-resolution inputs (every activation of the blocks): U pixels per thread and trip, all loads
    // issued before the first use; one block per CU then streams as fast as four with one-pixel trips
    constexpr int U = 4;
    const bool two = d.nin == 2;
    for (; p + (U - 1) * stride < npix; p += U * stride) {
      float f0[U][V], f1[U][V];
#pragma unroll
      for (int u = 0; u < U; ++u) VecIO<T, V>::ld(d.in[0].ptr, ((size_t)p + (size_t)u * stride) * d.in[0].pitch + c, f0[u]);
      if (two) {
#pragma unroll
        for (int u = 0; u < U; ++u) VecIO<T, V>::ld(d.in[1].ptr, ((size_t)p + (size_t)u * stride) * d.in[1].pitch + c, f1[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t pu = (size_t)p + (size_t)u * stride;
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = f0[u][i] * sc[0][i] + sh[0][i];
        if (two) {
#pragma unroll
          for (int i = 0; i < V; ++i) acc[i] += f1[u][i] * sc[1][i] + sh[1][i];
        }
        if (d.relu) {
          if (V > 1 && d.mask) {
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < V; ++i) bits |= (acc[i] > 0.f ? 1u : 0u) << i;
            d.mask[pu * d.mask_pitch + cv] = (uint8_t)bits;
          }
          if constexpr (!LEAKY) {
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] = fmaxf(acc[i], 0.f);
          } else {
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] = acc[i] > 0.f ? acc[i] : neg * acc[i];
          }
        }
        VecIO<T, V>::st(d.out, pu * d.out_pitch + c, acc);
      }
    }
  }
  for (; p < npix; p += stride) {
    unsigned x = 0, y = 0, n = 0;
    if (any_up) {
      const unsigned r = p / uW;
      x = p - r * uW;
      n = r / uH;
      y = r - n * uH;
    }
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXIN; ++j) {
      if (j >= d.nin) break;
      const hrp_ew_input& in = d.in[j];
      size_t q;
      if (in.up == 1) {
        q = (size_t)p * in.pitch + c;
      } else {
        const int lg = __ffs(in.up) - 1;   // up is 2, 4 or 8
        q = (((size_t)n * (uH >> lg) + (y >> lg)) * (uW >> lg) + (x >> lg)) * in.pitch + c;
      }
      float f[V];
      VecIO<T, V>::ld(in.ptr, q, f);
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] += f[i] * sc[j][i] + sh[j][i];
    }
    if (d.relu) {
      if (V > 1 && d.mask) {
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < V; ++i) bits |= (acc[i] > 0.f ? 1u : 0u) << i;
        d.mask[(size_t)p * d.mask_pitch + cv] = (uint8_t)bits;
      }
      if constexpr (!LEAKY) {
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = fmaxf(acc[i], 0.f);
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = acc[i] > 0.f ? acc[i] : neg * acc[i];
      }
    }
    VecIO<T, V>::st(d.out, (size_t)p * d.out_pitch + c, acc);
  }
}

template <typename T, int V, int MAXIN, bool LEAKY = false>
__global__ __launch_bounds__(256) void ew_fwd_kernel(const hrp_ew_desc d, int tpr, int nslab) {
  ew_fwd_body<T, V, MAXIN, LEAKY>(d, tpr, blockIdx.x, blockIdx.y, gridDim.x);
}

// pooled, masked output gradient of an upsampled fuse-layer input (up = UP: 2, 4 or 8) through the ReLU bit mask: a row of the
// UP x UP window at a time, its UP loads (and mask bytes) issued before the first use.  (The runtime-`up` loop below keeps one load
// in flight per thread: 38-41 us for the up = 8 inputs of a [64, 64, 64, 32] gradient, 0.45 TB/s.)  Same summation order.
template <typename T, int V, int UP>
__device__ __forceinline__ void pooled_grad_rows(const hrp_ew_bwd_desc& d, unsigned q, int c, float* g) {
  const unsigned Wq = d.W / UP, Hq = d.H / UP;
  const unsigned r = q / Wq, qx = q - r * Wq, n = r / Hq, qy = r - n * Hq;
  const size_t p0 = ((size_t)n * d.H + qy * UP) * d.W + qx * UP;
  for (int dy = 0; dy < UP; ++dy) {
    const size_t p = p0 + (size_t)dy * d.W;
    typename VecIO<T, V>::Raw raw[UP];
    unsigned bits[UP];
#pragma unroll
    for (int dx = 0; dx < UP; ++dx) raw[dx] = VecIO<T, V>::ldr(d.dout, (p + dx) * d.dout_pitch + c);
#pragma unroll
    for (int dx = 0; dx < UP; ++dx) bits[dx] = d.mask[(p + dx) * d.mask_pitch + c / V];
#pragma unroll
    for (int dx = 0; dx < UP; ++dx) {
      float go[V];
      VecIO<T, V>::cvt(raw[dx], go);
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] += (bits[dx] >> i) & 1u ? go[i] : 0.f;
    }
  }
}

// pooled, masked output gradient at input pixel q = (n, qy, qx)
template <typename T, int V, bool LEAKY = false>
__device__ __forceinline__ void pooled_grad(const hrp_ew_bwd_desc& d, unsigned q, int c, float* g) {
  if (d.pooled) {            // (uniform) the window sums were formed once for all terms of this activation: hrp_ew_pool2
    const float* q0 = d.pooled + (size_t)q * d.C + c;
#pragma unroll
    for (int i = 0; i < V; i += (V >= 4 ? 4 : 1)) {
      if constexpr (V >= 4) {
        const float4 t = *(const float4*)(q0 + i);
        g[i] = t.x; g[i + 1] = t.y; g[i + 2] = t.z; g[i + 3] = t.w;
      } else {
        g[i] = q0[i];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < V; ++i) g[i] = 0.f;
  const int up = d.in.up;
  if constexpr (V > 1 && !LEAKY) {
    if (up != 1 && d.relu == 1 && d.mask) {      // (uniform)
      if (up == 2) { pooled_grad_rows<T, V, 2>(d, q, c, g); return; }
      if (up == 4) { pooled_grad_rows<T, V, 4>(d, q, c, g); return; }
      if (up == 8) { pooled_grad_rows<T, V, 8>(d, q, c, g); return; }
    }
  }
  unsigned n = 0, qy = 0, qx = 0;
  if (up != 1) {   // (32-bit divisions, only for the upsampled inputs of the fuse layers)
    const unsigned Wq = d.W / up, Hq = d.H / up;
    const unsigned r = q / Wq;
    qx = q - r * Wq;
    n = r / Hq;
    qy = r - n * Hq;
  }
  for (int dy = 0; dy < up; ++dy)
    for (int dx = 0; dx < up; ++dx) {
      const size_t p = up == 1 ? (size_t)q : ((size_t)n * d.H + qy * up + dy) * d.W + qx * up + dx;
      float go[V];
      VecIO<T, V>::ld(d.dout, p * d.dout_pitch + c, go);
      if (d.relu) {
        if (V > 1 && d.mask) {
          const unsigned bits = d.mask[p * d.mask_pitch + c / V];
#pragma unroll
          for (int i = 0; i < V; ++i) go[i] = (bits >> i) & 1u ? go[i] : (LEAKY ? 0.01f * go[i] : 0.f);
        } else {
          float o[V];
          VecIO<T, V>::ld(d.out, p * d.out_pitch + c, o);
#pragma unroll
          for (int i = 0; i < V; ++i) go[i] = o[i] > 0.f ? go[i] : (LEAKY ? 0.01f * go[i] : 0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] += go[i];
    }
}

// Pixel loop of the reduce kernel for up == 1 (every activation except the upsampled fuse-layer inputs): U pixels per
// thread and trip, all loads issued before the first use.  One pixel per trip kept a single round of loads in flight
// per thread - 8 dependent HBM latencies on a [64,64,64,32] tensor, 1.6 TB/s.
// RM: 0 = no ReLU, 1 = ReLU through the bit mask, 2 = ReLU by comparing the saved output.
// LEAKY: relu == 2 (the slope multiply stays out of the ReLU instances: the reduce is VALU sensitive, +14 % measured)
template <typename T, int V, int RM, int U, bool LEAKY = false>
__device__ __forceinline__ unsigned reduce_pixels(const hrp_ew_bwd_desc& d, int c, unsigned q, unsigned stride, unsigned nq,
                                                  const float (&mean)[V], const float (&inv)[V], float (&s0)[V], float (&s1)[V]) {
  for (; q + (U - 1) * stride < nq; q += U * stride) {
    typename VecIO<T, V>::Raw go[U], xin[U];
    unsigned bits[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t p = (size_t)q + (size_t)u * stride;
      go[u] = VecIO<T, V>::ldr(d.dout, p * d.dout_pitch + c);
      xin[u] = VecIO<T, V>::ldr(d.in.ptr, p * d.in.pitch + c);
      if constexpr (RM == 1) bits[u] = d.mask[p * d.mask_pitch + c / V];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float gf[V], xf[V], o[V];   // (RM == 2, the rare variant without a bit mask: read where it is used)
      VecIO<T, V>::cvt(go[u], gf);
      VecIO<T, V>::cvt(xin[u], xf);
      if constexpr (RM == 2) VecIO<T, V>::ld(d.out, ((size_t)q + (size_t)u * stride) * d.out_pitch + c, o);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        float g = gf[i];
        if constexpr (RM == 1) g = (bits[u] >> i) & 1u ? g : (LEAKY ? 0.01f * g : 0.f);
        if constexpr (RM == 2) g = o[i] > 0.f ? g : (LEAKY ? 0.01f * g : 0.f);
        s0[i] += g;
        s1[i] += g * (xf[i] - mean[i]) * inv[i];
      }
    }
  }
  return q;
}

// 2 x 2 window sums of a gradient: thread = (output pixel, 8 channels).  SRC16: the source is bf16 (16 bytes per 8 channels), else fp32.
template <bool SRC16>
__global__ __launch_bounds__(256) void ew_pool2_kernel(const void* __restrict__ src, int src_pitch, const uint8_t* __restrict__ mask, int mask_pitch,
                                                       int N, int H, int W, int C, float* __restrict__ dst) {
  const int cv = C / 8, Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Ho * Wo * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c8 = (int)(i % cv);
    size_t r = i / cv;
    const int x = (int)(r % Wo);
    r /= Wo;
    const int y = (int)(r % Ho), n = (int)(r / Ho);
    const size_t p00 = ((size_t)n * H + 2 * y) * W + 2 * x;
    const size_t pp[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    float v[4][8];
    unsigned bits[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (SRC16) {
        const uint4 raw = *(const uint4*)((const char*)src + (pp[k] * src_pitch + 8 * c8) * 2);
        Elem<bf16_t>::unpack(raw, v[k]);
      } else {
        const float* q = (const float*)src + pp[k] * src_pitch + 8 * c8;
        const float4 a = *(const float4*)q, b = *(const float4*)(q + 4);
        v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z; v[k][3] = a.w; v[k][4] = b.x; v[k][5] = b.y; v[k][6] = b.z; v[k][7] = b.w;
      }
      // the mask has one byte per 16-byte vector of the SOURCE: 8 channels of bf16, 4 of fp32
      bits[k] = 0xffu;
      if (mask) bits[k] = SRC16 ? mask[pp[k] * mask_pitch + c8] : (unsigned)mask[pp[k] * mask_pitch + 2 * c8] | ((unsigned)mask[pp[k] * mask_pitch + 2 * c8 + 1] << 4);
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = (bits[0] >> e) & 1u ? v[0][e] : 0.f, b = (bits[1] >> e) & 1u ? v[1][e] : 0.f;
      const float c = (bits[2] >> e) & 1u ? v[2][e] : 0.f, dd = (bits[3] >> e) & 1u ? v[3][e] : 0.f;
      o[e] = (a + b) + (c + dd);
    }
    float* q = dst + i * 8;
    *(float4*)q = make_float4(o[0], o[1], o[2], o[3]);
    *(float4*)(q + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
}

template <typename T, int V>
__device__ __forceinline__ void ew_bwd_reduce_body(const hrp_ew_bwd_desc& d, const int tpr, const int bx, const int by, const int gx,
                                                   const int stat_slot) {
  extern __shared__ float ew_lds[];
  const int tabn = min(tpr * V, TAB_CH);
  float* tab = ew_lds;
  const int nred = (tpr < 64 ? 4 : 256 / tpr) * tpr * V;
  float* red0 = ew_lds + 4 * tabn;
  float* red1 = red0 + nred;
  const int lane_c = threadIdx.x % tpr;
  const int cv = by * tpr + lane_c;
  const int c = cv * V;
  const int cbase = by * tpr * V, nch = tpr * V;
  const int ppb = 256 / tpr;
  const int up = d.in.up, Hq = d.H / up, Wq = d.W / up;
  float s0[V], s1[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s0[i] = s1[i] = 0.f;
  float sc[V], sh[V], mean[V], inv[V];
  load_consts<V>(d.in, d.C, cbase, nch, c, tab, sc, sh, mean, inv);
  if (c < d.C) {
    const unsigned nq = (unsigned)d.N * Hq * Wq;
    const unsigned stride = gx * ppb;
    unsigned q = bx * ppb + threadIdx.x / tpr;
    if (up == 1) {   // (uniform) batched pixel loop, then its single-pixel tail
      if (!d.relu) { q = reduce_pixels<T, V, 0, 4>(d, c, q, stride, nq, mean, inv, s0, s1); q = reduce_pixels<T, V, 0, 1>(d, c, q, stride, nq, mean, inv, s0, s1); }
      else if (d.relu == 2) {   // LeakyReLU (rare: the add_fc MLP): one-pixel trips, mask or saved output
        if (V > 1 && d.mask) q = reduce_pixels<T, V, 1, 1, true>(d, c, q, stride, nq, mean, inv, s0, s1);
        else q = reduce_pixels<T, V, 2, 1, true>(d, c, q, stride, nq, mean, inv, s0, s1);
      }
      else if (V > 1 && d.mask) { q = reduce_pixels<T, V, 1, 4>(d, c, q, stride, nq, mean, inv, s0, s1); q = reduce_pixels<T, V, 1, 1>(d, c, q, stride, nq, mean, inv, s0, s1); }
      else { q = reduce_pixels<T, V, 2, 4>(d, c, q, stride, nq, mean, inv, s0, s1); q = reduce_pixels<T, V, 2, 1>(d, c, q, stride, nq, mean, inv, s0, s1); }
    }
    for (; q < nq; q += stride) {
      float g[V], xin[V];
      if (d.relu == 2) pooled_grad<T, V, true>(d, q, c, g);
      else pooled_grad<T, V>(d, q, c, g);
      VecIO<T, V>::ld(d.in.ptr, (size_t)q * d.in.pitch + c, xin);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        s0[i] += g[i];
        s1[i] += g[i] * (xin[i] - mean[i]) * inv[i];
      }
    }
  }
  // lanes lane_c, lane_c + tpr, ... of a wave hold the same channels: butterfly over the pixel lanes first
  if (tpr < 64) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      // inside a row of 16 lanes: DPP rotations (VALU speed); across rows: shuffles
      if (tpr <= 8) { s0[i] += ew_row_ror<8>(s0[i]); s1[i] += ew_row_ror<8>(s1[i]); }
      if (tpr <= 4) { s0[i] += ew_row_ror<4>(s0[i]); s1[i] += ew_row_ror<4>(s1[i]); }
      if (tpr <= 2) { s0[i] += ew_row_ror<2>(s0[i]); s1[i] += ew_row_ror<2>(s1[i]); }
      if (tpr <= 1) { s0[i] += ew_row_ror<1>(s0[i]); s1[i] += ew_row_ror<1>(s1[i]); }
      for (int o = 32; o >= (tpr > 16 ? tpr : 16); o >>= 1) {
        s0[i] += __shfl_xor(s0[i], o, 64);
        s1[i] += __shfl_xor(s1[i], o, 64);
      }
    }
  }
  // one partial per wave (tpr < 64) or per thread (tpr >= 64, then ppb = 256 / tpr threads share a channel)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nparts = tpr < 64 ? 4 : ppb;                 // partials per channel group in LDS
  const int part = tpr < 64 ? wave : threadIdx.x / tpr;  // which partial this thread writes
  const bool writer = tpr < 64 ? (lane < tpr) : true;
  if (writer) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      red0[(part * tpr + lane_c) * V + i] = s0[i];
      red1[(part * tpr + lane_c) * V + i] = s1[i];
    }
  }
  __syncthreads();
  // thread (lane_c, i) -> one channel: tpr * V threads finish the sum and issue the atomics
  for (int e = threadIdx.x; e < tpr * V; e += 256) {
    const int lc = e / V, i = e - lc * V;
    const int ch = (by * tpr + lc) * V + i;
    if (ch >= d.C) continue;
    float a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < nparts; ++k) {
      a0 += red0[(k * tpr + lc) * V + i];
      a1 += red1[(k * tpr + lc) * V + i];
    }
    double* slot = d.sums + stat_slot * 2 * d.C;
    atomicAdd(&slot[ch], (double)a0);
    atomicAdd(&slot[d.C + ch], (double)a1);
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void ew_bwd_reduce_kernel(const hrp_ew_bwd_desc d, int tpr) {
  ew_bwd_reduce_body<T, V>(d, tpr, blockIdx.x, blockIdx.y, gridDim.x, blockIdx.x & (HRP_STAT_SLOTS - 1));
}

#ifndef HRP_EW_APPLY_U
#define HRP_EW_APPLY_U 4
#endif
template <typename T, int V, bool LEAKY = false>
__device__ __forceinline__ void ew_bwd_apply_body(const hrp_ew_bwd_desc& d, const int tpr, const int bx, const int by, const int gx) {
  extern __shared__ float ew_lds[];
  const int tabn = min(tpr * V, TAB_CH);
  float* tab = ew_lds;
  float* ktab0 = ew_lds + 4 * tabn;
  float* ktab1 = ktab0 + tabn;
  const int cv = by * tpr + threadIdx.x % tpr;
  const int c = cv * V;
  const int cbase = by * tpr * V, nch = tpr * V;
  const int ppb = 256 / tpr;
  const int up = d.in.up, Hq = d.H / up, Wq = d.W / up;
  float sc[V], sh[V], mean[V], inv[V], k0[V], k1[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { sc[i] = 1.f; sh[i] = 0.f; mean[i] = 0.f; inv[i] = 1.f; k0[i] = k1[i] = 0.f; }
  if (d.in.mode != HRP_EW_IDENTITY) load_consts<V>(d.in, d.C, cbase, nch, c, tab, sc, sh, mean, inv);
  if (d.in.mode == HRP_EW_BN_TRAIN) {
    if (nch <= TAB_CH) {
      for (int e = threadIdx.x; e < nch; e += 256) {
        const int ch = cbase + e;
        ktab0[e] = ch < d.C ? slot_sum(d.sums, ch, 2 * d.C) / d.in.count : 0.f;
        ktab1[e] = ch < d.C ? slot_sum(d.sums, d.C + ch, 2 * d.C) / d.in.count : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int e = min(c - cbase + i, tabn - 1);
        k0[i] = ktab0[e]; k1[i] = ktab1[e];
      }
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i)
        if (c + i < d.C) {
          k0[i] = slot_sum(d.sums, c + i, 2 * d.C) / d.in.count;
          k1[i] = slot_sum(d.sums, d.C + c + i, 2 * d.C) / d.in.count;
        }
    }
  }
  if (c >= d.C) return;
  const unsigned nq = (unsigned)d.N * Hq * Wq;
  const unsigned stride = gx * ppb;
  unsigned q = bx * ppb + threadIdx.x / tpr;
  if (up == 1) {
    // U pixels per thread and trip, every load of the trip issued before the first use (the branches are uniform)
    constexpr int U = HRP_EW_APPLY_U;
    const bool bn = d.in.mode == HRP_EW_BN_TRAIN, use_bits = d.relu && V > 1 && d.mask, use_out = d.relu && !use_bits;
    constexpr float neg = LEAKY ? 0.01f : 0.f;
    for (; q + (U - 1) * stride < nq; q += U * stride) {
      // (the rarer operands - saved output instead of the bit mask, accumulation targets - are read where they are
      // used: keeping them in the batch cost 100 registers on every variant of the kernel)
      float g[U][V], xin[U][V];
      unsigned bits[U];
#pragma unroll
      for (int u = 0; u < U; ++u) VecIO<T, V>::ld(d.dout, ((size_t)q + (size_t)u * stride) * d.dout_pitch + c, g[u]);
      if (bn) {
#pragma unroll
        for (int u = 0; u < U; ++u) VecIO<T, V>::ld(d.in.ptr, ((size_t)q + (size_t)u * stride) * d.in.pitch + c, xin[u]);
      }
      if (use_bits) {
#pragma unroll
        for (int u = 0; u < U; ++u) bits[u] = d.mask[((size_t)q + (size_t)u * stride) * d.mask_pitch + c / V];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t p = (size_t)q + (size_t)u * stride;
        if (use_bits) {
#pragma unroll
          for (int i = 0; i < V; ++i) g[u][i] = (bits[u] >> i) & 1u ? g[u][i] : (LEAKY ? neg * g[u][i] : 0.f);
        } else if (use_out) {
          float o[V];
          VecIO<T, V>::ld(d.out, p * d.out_pitch + c, o);
#pragma unroll
          for (int i = 0; i < V; ++i) g[u][i] = o[i] > 0.f ? g[u][i] : (LEAKY ? neg * g[u][i] : 0.f);
        }
        if (d.din2) {
          if (d.accumulate2) {
            float g2[V];
            VecIO<T, V>::ld(d.din2, p * d.din2_pitch + c, g2);
#pragma unroll
            for (int i = 0; i < V; ++i) g2[i] += g[u][i];
            VecIO<T, V>::st(d.din2, p * d.din2_pitch + c, g2);
          } else {
            VecIO<T, V>::st(d.din2, p * d.din2_pitch + c, g[u]);
          }
        }
        if (bn) {
#pragma unroll
          for (int i = 0; i < V; ++i) g[u][i] = sc[i] * (g[u][i] - k0[i] - (xin[u][i] - mean[i]) * inv[i] * k1[i]);
        } else if (d.in.mode == HRP_EW_AFFINE) {
#pragma unroll
          for (int i = 0; i < V; ++i) g[u][i] *= sc[i];
        }
        if (d.accumulate) {
          float old[V];
          VecIO<T, V>::ld(d.din, p * d.din_pitch + c, old);
#pragma unroll
          for (int i = 0; i < V; ++i) g[u][i] += old[i];
        }
        VecIO<T, V>::st(d.din, p * d.din_pitch + c, g[u]);
      }
    }
  }
  for (; q < nq; q += stride) {
    float g[V];
    pooled_grad<T, V, LEAKY>(d, q, c, g);
    if (d.din2) {   // identity sibling of the same activation (up == 1): its gradient is g itself
      float g2[V];
      const size_t o2 = (size_t)q * d.din2_pitch + c;
      if (d.accumulate2) {
        VecIO<T, V>::ld(d.din2, o2, g2);
#pragma unroll
        for (int i = 0; i < V; ++i) g2[i] += g[i];
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) g2[i] = g[i];
      }
      VecIO<T, V>::st(d.din2, o2, g2);
    }
    if (d.in.mode == HRP_EW_BN_TRAIN) {
      float xin[V];
      VecIO<T, V>::ld(d.in.ptr, (size_t)q * d.in.pitch + c, xin);
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] = sc[i] * (g[i] - k0[i] - (xin[i] - mean[i]) * inv[i] * k1[i]);
    } else if (d.in.mode == HRP_EW_AFFINE) {
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] *= sc[i];
    }
    size_t o = (size_t)q * d.din_pitch + c;
    if (d.accumulate) {
      float old[V];
      VecIO<T, V>::ld(d.din, o, old);
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] += old[i];
    }
    VecIO<T, V>::st(d.din, o, g);
  }
}

template <typename T, int V, bool LEAKY = false>
__global__ __launch_bounds__(256) void ew_bwd_apply_kernel(const hrp_ew_bwd_desc d, int tpr) {
  ew_bwd_apply_body<T, V, LEAKY>(d, tpr, blockIdx.x, blockIdx.y, gridDim.x);
}

static inline bool aligned16(const void* p, int pitch, int sz) {
  return ((uintptr_t)p % 16 == 0) && (((size_t)pitch * sz) % 16 == 0);
}

struct EwGeom { int V, tpr, nslab, gx; };

// maxblocks: 4 workgroups per CU for the streaming kernels; the reduce kernel uses fewer, each block ends
// with 2 * channels atomics
static EwGeom geom(int C, int vec, bool vec_ok, long npix, int maxblocks) {
  EwGeom g;
  g.V = (vec_ok && C % vec == 0) ? vec : 1;
  int nv = C / g.V;
  int tpr = 1;
  // at most 64 vector columns (512 channels) per block: the block's channel constants then always go through the
  // LDS table (per-thread derivation = 16 dependent-latency loads per channel, 10x slower on the 1024 / 2048-channel
  // ResNet layers); wider tensors are split into channel slabs (grid.y)
  const int tpr_max = vec_ok && C % vec == 0 ? TAB_CH / vec : 256;
  while (tpr < nv && tpr < tpr_max && tpr < 256) tpr <<= 1;
  g.tpr = tpr;
  g.nslab = cdiv(nv, tpr);
  int ppb = 256 / tpr;
  long blocks = (npix + ppb - 1) / ppb;
  static const int mult = 1;   // (swept: DESIGN 5)
  long cap = (long)maxblocks * mult / g.nslab;
  if (cap < 1) cap = 1;
  g.gx = (int)(blocks < cap ? blocks : cap);
  if (g.gx < 1) g.gx = 1;
  return g;
}

template <typename T>
static int ew_fwd_t(const hrp_ew_desc& d, hipStream_t s) {
  constexpr int SZ = Elem<T>::SZ, VEC = Elem<T>::VEC;
  bool ok = aligned16(d.out, d.out_pitch, SZ);
  for (int j = 0; j < d.nin; ++j) ok = ok && aligned16(d.in[j].ptr, d.in[j].pitch, SZ);
  // (one block per CU with 4 pixels in flight per thread: 47.1 ms per step at 1 024 blocks, 46.8 at 512, 46.4 at 256)
  static const int fwd_blocks = 256;   // (swept: DESIGN 5)
  bool batched = d.nin <= 2;   // the kernel's four-pixel path; fuse sums (3-4 inputs, upsampled ones) keep 4 blocks per CU
  for (int j = 0; j < d.nin; ++j) batched = batched && d.in[j].up == 1;
  EwGeom g = geom(d.C, VEC, ok, (long)d.N * d.H * d.W, batched ? fwd_blocks : 1024);
  HRP_REQUIRE(!d.mask || (d.relu && g.V == VEC && d.mask_pitch >= d.C / VEC), "ew_fwd: the ReLU bit mask needs relu and the 16-byte vector path");
  dim3 grid(g.gx, g.nslab);
  const int tabn = g.tpr * g.V < TAB_CH ? g.tpr * g.V : TAB_CH;
  const int lds = 4 * tabn * 4;
  if (d.relu == 2) {      // LeakyReLU: its own instances (rare)
    if (g.V == 1) hipLaunchKernelGGL((ew_fwd_kernel<T, 1, HRP_EW_MAX_IN, true>), grid, dim3(256), lds, s, d, g.tpr, g.nslab);
    else hipLaunchKernelGGL((ew_fwd_kernel<T, VEC, HRP_EW_MAX_IN, true>), grid, dim3(256), lds, s, d, g.tpr, g.nslab);
    return check_launch("ew_fwd");
  }
  if (g.V == 1) hipLaunchKernelGGL((ew_fwd_kernel<T, 1, HRP_EW_MAX_IN>), grid, dim3(256), lds, s, d, g.tpr, g.nslab);
  else if (d.nin <= 2) hipLaunchKernelGGL((ew_fwd_kernel<T, VEC, 2>), grid, dim3(256), lds, s, d, g.tpr, g.nslab);
  else hipLaunchKernelGGL((ew_fwd_kernel<T, VEC, HRP_EW_MAX_IN>), grid, dim3(256), lds, s, d, g.tpr, g.nslab);
  return check_launch("ew_fwd");
}

template <typename T, bool APPLY>
static int ew_bwd_t(const hrp_ew_bwd_desc& d, hipStream_t s) {
  constexpr int SZ = Elem<T>::SZ, VEC = Elem<T>::VEC;
  bool ok = aligned16(d.dout, d.dout_pitch, SZ) && aligned16(d.in.ptr ? d.in.ptr : d.dout, d.in.pitch ? d.in.pitch : d.dout_pitch, SZ);
  if (d.relu) ok = ok && aligned16(d.out, d.out_pitch, SZ);
  if (APPLY) ok = ok && aligned16(d.din, d.din_pitch, SZ);
  if (APPLY && d.din2) ok = ok && aligned16(d.din2, d.din2_pitch, SZ);
  const int up = d.in.up;
  static const int red_blocks = 256;   // (swept: DESIGN 5)
  // (the reduce ends with 2 C atomics per block and keeps 4 pixels of loads in flight per thread: one block per CU
  // streams as fast alone as 512 and leaves the CUs to the kernels of the other lanes - 48.3 -> 47.6 ms per step;
  // twice as many on the >= 64 MiB tensors, where the streaming part dominates)
  const bool big = (int64_t)d.N * d.H * d.W * d.C * SZ >= (64ll << 20);
  // (apply: 4 pixels of loads in flight per thread and ONE block per CU - 1 024 blocks of one-pixel trips took 9.2 ms
  // of kernel time per step and a 47.6 ms step, this 9.8 ms and 46.8 ms: the CUs stay free for the other lanes;
  // block counts that are not a multiple of the 256 CUs (192, 320, 384) lose 0.5-1 ms to the uneven tail)
  static const int apply_blocks = 256;   // (swept: DESIGN 5)
  // (the upsampled fuse-layer inputs keep the one-pixel loop: they get the block counts that suited it)
  const int nblk = up != 1 ? (APPLY ? 1024 : 512) : APPLY ? apply_blocks : (big ? 2 * red_blocks : red_blocks);
  EwGeom g = geom(d.C, VEC, ok, (long)d.N * (d.H / up) * (d.W / up), nblk);
  HRP_REQUIRE(!d.mask || (d.relu && g.V == VEC && d.mask_pitch >= d.C / VEC), "ew_bwd: the ReLU bit mask needs relu and the 16-byte vector path");
  dim3 grid(g.gx, g.nslab);
  const int tabn = g.tpr * g.V < TAB_CH ? g.tpr * g.V : TAB_CH;
  if (APPLY) {
    const int lds = 6 * tabn * 4;
    if (d.relu == 2) {
      if (g.V == 1) hipLaunchKernelGGL((ew_bwd_apply_kernel<T, 1, true>), grid, dim3(256), lds, s, d, g.tpr);
      else hipLaunchKernelGGL((ew_bwd_apply_kernel<T, VEC, true>), grid, dim3(256), lds, s, d, g.tpr);
    } else if (g.V == 1) hipLaunchKernelGGL((ew_bwd_apply_kernel<T, 1>), grid, dim3(256), lds, s, d, g.tpr);
    else hipLaunchKernelGGL((ew_bwd_apply_kernel<T, VEC>), grid, dim3(256), lds, s, d, g.tpr);
  } else {
    const int nred = (g.tpr < 64 ? 4 : 256 / g.tpr) * g.tpr * g.V;
    const int lds = (4 * tabn + 2 * nred) * 4;
    if (g.V == 1) hipLaunchKernelGGL((ew_bwd_reduce_kernel<T, 1>), grid, dim3(256), lds, s, d, g.tpr);
    else hipLaunchKernelGGL((ew_bwd_reduce_kernel<T, VEC>), grid, dim3(256), lds, s, d, g.tpr);
  }
  return check_launch(APPLY ? "ew_bwd_apply" : "ew_bwd_reduce");
}


// ---- batched launches (hrp_batch_*, include/hrp.h) --------------------------------------------------------------
// n element-wise problems in one launch (the activations of every branch of both trunks after the same layer, all
// BatchNorm backward passes of a lock-step layer ..).  Vector path only.  The launch's workgroups are shared between
// the problems in proportion to their bytes, so a workgroup of the [64,64,64,32] tensor and one of the [64,8,8,256]
// tensor stream about the same amount.
struct EwProblem { hrp_ew_desc d; int tpr, gx, nslab, pad; FastDiv fd_gx; };
struct EwBwdProblem { hrp_ew_bwd_desc d; int tpr, gx, nslab, pad; FastDiv fd_gx; };

template <typename T, int MAXIN>
__global__ __launch_bounds__(256) void ew_fwd_batch_kernel(const EwProblem* __restrict__ tab, const BatchHdr h) {
  int base;
  const int g = batch_find(h, blockIdx.x, base);
  const EwProblem& P = tab[g];
  const int local = (int)blockIdx.x - base;
  const int by = fdiv(local, P.fd_gx), bx = local - by * P.gx;
  ew_fwd_body<T, Elem<T>::VEC, MAXIN>(P.d, P.tpr, bx, by, P.gx);
}

template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void ew_bwd_batch_kernel(const EwBwdProblem* __restrict__ tab, const BatchHdr h) {
  int base;
  const int g = batch_find(h, blockIdx.x, base);
  const EwBwdProblem& P = tab[g];
  const int local = (int)blockIdx.x - base;
  const int by = fdiv(local, P.fd_gx), bx = local - by * P.gx;
  if constexpr (APPLY) ew_bwd_apply_body<T, Elem<T>::VEC>(P.d, P.tpr, bx, by, P.gx);
  else ew_bwd_reduce_body<T, Elem<T>::VEC>(P.d, P.tpr, bx, by, P.gx, blockIdx.x & (HRP_STAT_SLOTS - 1));
}

static int ew_fwd_check(const hrp_ew_desc* d);
static int ew_bwd_check(const hrp_ew_bwd_desc* d, bool apply);

// share of `total` workgroups for a problem of `bytes` out of `sum` (at least `lo`)
static inline int ew_share(int total, double bytes, double sum, int lo) {
  int b = (int)(total * bytes / sum + 0.5);
  return b < lo ? lo : b;
}

template <typename T>
static int ew_fwd_batch_prepare(const hrp_ew_desc* descs, int n, EwProblem* tab, hrp_batch_info* info) {
  constexpr int SZ = Elem<T>::SZ, VEC = Elem<T>::VEC;
  double bytes[HRP_BATCH_MAX], sum = 0.0;
  int maxin = 2;
  for (int i = 0; i < n; ++i) {
    const hrp_ew_desc& d = descs[i];
    const int rc = ew_fwd_check(&d);
    if (rc != HRP_OK) return rc;
    bool ok = aligned16(d.out, d.out_pitch, SZ) && d.C % VEC == 0;
    for (int j = 0; j < d.nin; ++j) ok = ok && aligned16(d.in[j].ptr, d.in[j].pitch, SZ);
    HRP_REQUIRE(ok, "ew batch: problem %d is not on the 16-byte vector path", i);
    HRP_REQUIRE(d.relu != 2, "ew batch: LeakyReLU problems are launched one by one");
    HRP_REQUIRE(!d.mask || (d.relu && d.mask_pitch >= d.C / VEC), "ew_fwd: the ReLU bit mask needs relu and the 16-byte vector path");
    if (d.nin > 2) maxin = HRP_EW_MAX_IN;
    bytes[i] = (double)d.N * d.H * d.W * d.C * SZ;
    sum += bytes[i];
  }
  // (block budgets swept 256 .. 8192 on the B=64 step: 768 / 384 is the minimum - 43.1 ms at 2048 / 1024, 41.0-41.8 here;
  // issuing the first trip's loads before the channel-constant phase was measured 0.5 ms slower)
  static const int total = 768;
  int blk = 0, lds_max = 0;
  for (int i = 0; i < n; ++i) {
    const hrp_ew_desc& d = descs[i];
    const EwGeom g = geom(d.C, VEC, true, (long)d.N * d.H * d.W, ew_share(n == 1 ? 256 : total, bytes[i], sum, 16));
    const int tabn = g.tpr * g.V < TAB_CH ? g.tpr * g.V : TAB_CH;
    lds_max = 4 * tabn * 4 > lds_max ? 4 * tabn * 4 : lds_max;
    info->blk0[i] = blk;
    blk += g.gx * g.nslab;
    if (tab) {
      memset(&tab[i], 0, sizeof(EwProblem));
      tab[i].d = d; tab[i].tpr = g.tpr; tab[i].gx = g.gx; tab[i].nslab = g.nslab; tab[i].fd_gx = make_fastdiv(g.gx);
    }
  }
  info->blk0[n] = blk;
  info->grid = blk; info->lds_bytes = lds_max; info->variant = maxin;
  return HRP_OK;
}

template <typename T, bool APPLY>
static int ew_bwd_batch_prepare(const hrp_ew_bwd_desc* descs, int n, EwBwdProblem* tab, hrp_batch_info* info) {
  constexpr int SZ = Elem<T>::SZ, VEC = Elem<T>::VEC;
  double bytes[HRP_BATCH_MAX], sum = 0.0;
  for (int i = 0; i < n; ++i) {
    const hrp_ew_bwd_desc& d = descs[i];
    const int rc = ew_bwd_check(&d, APPLY);
    if (rc != HRP_OK) return rc;
    bool ok = aligned16(d.dout, d.dout_pitch, SZ) && aligned16(d.in.ptr ? d.in.ptr : d.dout, d.in.pitch ? d.in.pitch : d.dout_pitch, SZ) &&
              d.C % VEC == 0;
    if (d.relu) ok = ok && aligned16(d.out, d.out_pitch, SZ);
    if (APPLY) ok = ok && aligned16(d.din, d.din_pitch, SZ);
    if (APPLY && d.din2) ok = ok && aligned16(d.din2, d.din2_pitch, SZ);
    HRP_REQUIRE(ok, "ew batch: problem %d is not on the 16-byte vector path", i);
    HRP_REQUIRE(d.relu != 2, "ew batch: LeakyReLU problems are launched one by one");
    HRP_REQUIRE(!d.mask || (d.relu && d.mask_pitch >= d.C / VEC), "ew_bwd: the ReLU bit mask needs relu and the 16-byte vector path");
    bytes[i] = (double)d.N * d.H * d.W * d.C * SZ;
    sum += bytes[i];
  }
  static const int total_apply = 768;
  static const int total_red = 384;
  int blk = 0, lds_max = 0;
  for (int i = 0; i < n; ++i) {
    const hrp_ew_bwd_desc& d = descs[i];
    const int up = d.in.up;
    const int share = ew_share(n == 1 ? 256 : (APPLY ? total_apply : total_red), bytes[i], sum, 16);
    const EwGeom g = geom(d.C, VEC, true, (long)d.N * (d.H / up) * (d.W / up), share);
    const int tabn = g.tpr * g.V < TAB_CH ? g.tpr * g.V : TAB_CH;
    int lds;
    if (APPLY) lds = 6 * tabn * 4;
    else lds = (4 * tabn + 2 * ((g.tpr < 64 ? 4 : 256 / g.tpr) * g.tpr * g.V)) * 4;
    lds_max = lds > lds_max ? lds : lds_max;
    info->blk0[i] = blk;
    blk += g.gx * g.nslab;
    if (tab) {
      memset(&tab[i], 0, sizeof(EwBwdProblem));
      tab[i].d = d; tab[i].tpr = g.tpr; tab[i].gx = g.gx; tab[i].nslab = g.nslab; tab[i].fd_gx = make_fastdiv(g.gx);
    }
  }
  info->blk0[n] = blk;
  info->grid = blk; info->lds_bytes = lds_max; info->variant = 0;
  return HRP_OK;
}

int ew_batch_prepare(int family, const void* descs, int n, void* table, hrp_batch_info* info) {
  if (family == HRP_BATCH_EW_FWD) {
    const hrp_ew_desc* d = (const hrp_ew_desc*)descs;
    for (int i = 0; i < n; ++i) HRP_REQUIRE(d[i].dtype == d[0].dtype, "ew batch: mixed element types");
    info->dtype = d[0].dtype;
    return d[0].dtype == HRP_F32 ? ew_fwd_batch_prepare<float>(d, n, (EwProblem*)table, info)
                                 : ew_fwd_batch_prepare<bf16_t>(d, n, (EwProblem*)table, info);
  }
  const hrp_ew_bwd_desc* d = (const hrp_ew_bwd_desc*)descs;
  for (int i = 0; i < n; ++i) HRP_REQUIRE(d[i].dtype == d[0].dtype, "ew batch: mixed element types");
  info->dtype = d[0].dtype;
  const bool f32 = d[0].dtype == HRP_F32;
  if (family == HRP_BATCH_EW_BWD_APPLY)
    return f32 ? ew_bwd_batch_prepare<float, true>(d, n, (EwBwdProblem*)table, info) : ew_bwd_batch_prepare<bf16_t, true>(d, n, (EwBwdProblem*)table, info);
  return f32 ? ew_bwd_batch_prepare<float, false>(d, n, (EwBwdProblem*)table, info) : ew_bwd_batch_prepare<bf16_t, false>(d, n, (EwBwdProblem*)table, info);
}

int ew_batch_launch(const void* table_dev, const hrp_batch_info* info, hipStream_t s) {
  const BatchHdr h = make_hdr(info->blk0, info->n);
  const dim3 grid(info->grid), block(256);
  const bool f32 = info->dtype == HRP_F32;
  if (info->family == HRP_BATCH_EW_FWD) {
    const EwProblem* tab = (const EwProblem*)table_dev;
    if (info->variant <= 2) {
      if (f32) hipLaunchKernelGGL((ew_fwd_batch_kernel<float, 2>), grid, block, info->lds_bytes, s, tab, h);
      else hipLaunchKernelGGL((ew_fwd_batch_kernel<bf16_t, 2>), grid, block, info->lds_bytes, s, tab, h);
    } else {
      if (f32) hipLaunchKernelGGL((ew_fwd_batch_kernel<float, HRP_EW_MAX_IN>), grid, block, info->lds_bytes, s, tab, h);
      else hipLaunchKernelGGL((ew_fwd_batch_kernel<bf16_t, HRP_EW_MAX_IN>), grid, block, info->lds_bytes, s, tab, h);
    }
    return check_launch("ew_fwd_batch_kernel");
  }
  const EwBwdProblem* tab = (const EwBwdProblem*)table_dev;
  if (info->family == HRP_BATCH_EW_BWD_APPLY) {
    if (f32) hipLaunchKernelGGL((ew_bwd_batch_kernel<float, true>), grid, block, info->lds_bytes, s, tab, h);
    else hipLaunchKernelGGL((ew_bwd_batch_kernel<bf16_t, true>), grid, block, info->lds_bytes, s, tab, h);
  } else {
    if (f32) hipLaunchKernelGGL((ew_bwd_batch_kernel<float, false>), grid, block, info->lds_bytes, s, tab, h);
    else hipLaunchKernelGGL((ew_bwd_batch_kernel<bf16_t, false>), grid, block, info->lds_bytes, s, tab, h);
  }
  return check_launch("ew_bwd_batch_kernel");
}

int64_t ew_batch_table_bytes(int family, int n) {
  return (int64_t)n * (family == HRP_BATCH_EW_FWD ? sizeof(EwProblem) : sizeof(EwBwdProblem));
}

}  // namespace hrp

using namespace hrp;

static int hrp::ew_fwd_check(const hrp_ew_desc* d) {
  HRP_REQUIRE(d && d->out && d->nin >= 1 && d->nin <= HRP_EW_MAX_IN, "ew_fwd: bad descriptor");
  HRP_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0, "ew_fwd: empty");
  HRP_REQUIRE((int64_t)d->N * d->H * d->W < (1ll << 31), "ew_fwd: more than 2^31 pixels");
  for (int j = 0; j < d->nin; ++j) {
    const hrp_ew_input& in = d->in[j];
    HRP_REQUIRE(in.ptr && in.up >= 1 && d->H % in.up == 0 && d->W % in.up == 0, "ew_fwd: input %d geometry", j);
    HRP_REQUIRE(in.mode == HRP_EW_IDENTITY || (in.a && in.b), "ew_fwd: input %d needs a/b", j);
    HRP_REQUIRE(in.mode != HRP_EW_BN_TRAIN || (in.stats && in.count > 0.f), "ew_fwd: input %d needs stats", j);
  }
  HRP_REQUIRE(!d->consts_out || d->in[0].mode == HRP_EW_BN_TRAIN, "ew_fwd: consts_out needs a train-mode BatchNorm as input 0");
  return HRP_OK;
}

extern "C" int hrp_ew_fwd(const hrp_ew_desc* d, void* stream) {
  const int rc = ew_fwd_check(d);
  if (rc) return rc;
  if (d->dtype == HRP_F32) return ew_fwd_t<float>(*d, (hipStream_t)stream);
  return ew_fwd_t<bf16_t>(*d, (hipStream_t)stream);
}

static int hrp::ew_bwd_check(const hrp_ew_bwd_desc* d, bool apply) {
  HRP_REQUIRE(d && d->dout, "ew_bwd: bad descriptor");
  HRP_REQUIRE(!d->relu || d->out, "ew_bwd: relu needs the forward output");
  HRP_REQUIRE(d->in.up >= 1 && d->H % d->in.up == 0 && d->W % d->in.up == 0, "ew_bwd: geometry");
  HRP_REQUIRE((int64_t)d->N * d->H * d->W < (1ll << 31), "ew_bwd: more than 2^31 pixels");
  HRP_REQUIRE(d->in.mode == HRP_EW_IDENTITY || d->in.ptr, "ew_bwd: needs forward input values");
  HRP_REQUIRE(d->in.mode != HRP_EW_BN_TRAIN || (d->sums && d->in.stats && d->in.a), "ew_bwd: bn needs sums/stats");
  HRP_REQUIRE(!apply || d->din, "ew_bwd_apply: din");
  HRP_REQUIRE(!d->din2 || d->in.up == 1, "ew_bwd: din2 needs up == 1");   /* (the reduce pass ignores din2) */
  HRP_REQUIRE(!d->pooled || (d->in.up > 1 && d->C % 8 == 0 && (uintptr_t)d->pooled % 16 == 0), "ew_bwd: pooled needs up > 1, C %% 8 == 0, 16-byte alignment");
  HRP_REQUIRE(apply || (d->sums && d->in.ptr), "ew_bwd_reduce: sums / input values");
  return HRP_OK;
}

extern "C" int hrp_ew_bwd_reduce(const hrp_ew_bwd_desc* d, void* stream) {
  int rc = ew_bwd_check(d, false);
  if (rc) return rc;
  if (d->dtype == HRP_F32) return ew_bwd_t<float, false>(*d, (hipStream_t)stream);
  return ew_bwd_t<bf16_t, false>(*d, (hipStream_t)stream);
}

extern "C" int hrp_ew_pool2(const void* src, int src_dtype, int src_pitch, const uint8_t* mask, int mask_pitch, int N, int H, int W, int C,
                            float* dst, void* stream) {
  HRP_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0 && src_pitch >= C,
              "ew_pool2: geometry (H, W even, C a multiple of 8)");
  HRP_REQUIRE(src_dtype == HRP_BF16 || src_dtype == HRP_F32, "ew_pool2: source dtype");
  const int esz = src_dtype == HRP_BF16 ? 2 : 4;
  HRP_REQUIRE((uintptr_t)src % 16 == 0 && ((size_t)src_pitch * esz) % 16 == 0 && (uintptr_t)dst % 16 == 0, "ew_pool2: alignment");
  HRP_REQUIRE(!mask || mask_pitch >= C / (16 / esz), "ew_pool2: mask pitch");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 8);
  const unsigned blocks = (unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  if (src_dtype == HRP_BF16) hipLaunchKernelGGL((ew_pool2_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, src_pitch, mask, mask_pitch, N, H, W, C, dst);
  else hipLaunchKernelGGL((ew_pool2_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, src_pitch, mask, mask_pitch, N, H, W, C, dst);
  return check_launch("ew_pool2");
}

extern "C" int hrp_ew_bwd_apply(const hrp_ew_bwd_desc* d, void* stream) {
  int rc = ew_bwd_check(d, true);
  if (rc) return rc;
  if (d->dtype == HRP_F32) return ew_bwd_t<float, true>(*d, (hipStream_t)stream);
  return ew_bwd_t<bf16_t, true>(*d, (hipStream_t)stream);
}
