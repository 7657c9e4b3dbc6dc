// Convolution weight gradient on MFMA for gfx950.
//
//   dW[co][ci][t] (+)= sum over output pixels p of dY[p][co] * X[p shifted by tap t][ci]
//
// GEMM view per tap: M = cout, N = cin, K = pixels (the long dimension).  A workgroup owns one
// 32-cout x 32-cin block for ALL taps and walks a strided share of the pixel tiles; per tile it stages
// the dY tile and the X halo tile (32 channels each) in LDS once and every tap re-reads the X tile at a
// shifted offset.  The 4 waves split the tile's pixels (split-K inside the workgroup), partial sums are
// combined through LDS and leave as fp32 atomics straight into the PyTorch-shaped gradient tensor.
//
// Operand gather: the reduction index is the pixel, but NHWC keeps channels contiguous, so the 8 k-values
// a lane needs for the bf16 MFMA belong to 8 different pixels.  gfx950's LDS transpose read
// (ds_read_b64_tr_b16) does exactly that re-layout: within a 16-lane group source lane 4j+t supplies 4
// contiguous channels (8 bytes) of pixel j and destination lane i receives channel i of pixels 0..3
// (mapping measured with tools/probe_tr.hip: result[i][j] = src[4j + (i >> 2)][i & 3]).  Two such reads
// build one 8-deep MFMA operand; every source lane carries its own pixel address, so tap shifts and
// stride-2 sampling cost nothing extra.  fp32 needs one 32-bit read per operand.
#include "hrp_common.h"

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
  int lds_dy_off, lds_tab_off, lds_red_off;
  int use_ws, lds_bytes;
  FastDiv fd_ihw, fd_iwt, fd_thw, fd_tw, fd_tx, fd_ty, fd_cib;
};

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

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const hrp_wgrad_desc d, const WgradTiling t) {
  constexpr int SZ = Elem<T>::SZ, VEC = Elem<T>::VEC;
  constexpr int ROWB = 32 * SZ;     // bytes of 32 channels
  constexpr int P = ROWB + 16;      // LDS pixel pitch
  constexpr int NVEC = ROWB / 16;   // 16-byte vectors per pixel row
  constexpr int K = WG<T>::K;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_x = smem;
  char* lds_dy = smem + t.lds_dy_off;
  int* xtab = (int*)(smem + t.lds_tab_off);
  float* red = (float*)(smem + t.lds_red_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int blk = blockIdx.y;
  const int cob = fdiv(blk, t.fd_cib), cib = blk - cob * t.n_cib;
  const int co0 = cob * 32, ci0 = cib * 32;
  const int IS = d.in_stride;
  const int thw = t.TH * t.TW, ihw = t.IHt * t.IWt;

  // pixel -> X-tile byte offset table
  for (int m = tid; m < t.BM; m += 256) {
    int ti = fdiv(m, t.fd_thw), rem = m - ti * thw;
    int ty = fdiv(rem, t.fd_tw), tx = rem - ty * t.TW;
    if (ti >= t.TI) ti = t.TI - 1;  // idle slot (its dY row is zero)
    xtab[m] = ((ti * t.IHt + ty * IS) * t.IWt + tx * IS) * P;
  }

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  const char* xg = (const char*)d.x;
  const char* dyg = (const char*)d.dy;
  const int ppw = t.BM / 4;  // pixels per wave
  const int x_vecs = t.in_pix * NVEC, dy_vecs = t.BM * NVEC;

  int tapoff[NT];
#pragma unroll
  for (int tp = 0; tp < NT; ++tp) tapoff[tp] = ((d.dy_t[tp] - t.mindy) * t.IWt + (d.dx_t[tp] - t.mindx)) * P;

  // transpose-read lane roles (bf16): source lane s = lane & 15 -> pixel (s >> 2) of the 4-pixel block,
  // channels rbase + 4 (s & 3) .. +3; rbase = 16 for the odd 16-lane groups
  const int tr_pix = (lane & 15) >> 2;
  const int tr_coff = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * SZ;

  for (int tile = blockIdx.x; tile < t.ntiles; tile += t.G) {
    int q = fdiv(tile, t.fd_tx);
    const int tx_i = tile - q * t.tiles_x;
    const int tn_i = fdiv(q, t.fd_ty);
    const int ty_i = q - tn_i * t.tiles_y;
    const int n0 = tn_i * t.TI, oy0 = ty_i * t.TH, ox0 = tx_i * t.TW;
    const int iy0 = oy0 * IS + t.mindy, ix0 = ox0 * IS + t.mindx;
    __syncthreads();
    for (int v0 = tid; v0 < x_vecs; v0 += 256 * WSTAGE_U) {
      uint4 val[WSTAGE_U];
      int dst[WSTAGE_U];
#pragma unroll
      for (int u = 0; u < WSTAGE_U; ++u) {
        const int v = v0 + u * 256;
        val[u] = make_uint4(0, 0, 0, 0);
        dst[u] = -1;
        if (v < x_vecs) {
          int pix = v / NVEC, vec = v - pix * NVEC;
          int ti = fdiv(pix, t.fd_ihw), rem = pix - ti * ihw;
          int iy = fdiv(rem, t.fd_iwt), ix = rem - iy * t.IWt;
          int n = n0 + ti, gy = iy0 + iy, gx = ix0 + ix;
          int c = ci0 + vec * VEC;
          dst[u] = pix * P + vec * 16;
          if (n < d.N && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W && c < d.Cin) {
            size_t off = (((size_t)n * d.H + gy) * d.W + gx) * (size_t)d.x_pitch + c;
            val[u] = *(const uint4*)(xg + off * SZ);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < WSTAGE_U; ++u)
        if (dst[u] >= 0) *(uint4*)(lds_x + dst[u]) = val[u];
    }
    for (int v0 = tid; v0 < dy_vecs; v0 += 256 * WSTAGE_U) {
      uint4 val[WSTAGE_U];
      int dst[WSTAGE_U];
#pragma unroll
      for (int u = 0; u < WSTAGE_U; ++u) {
        const int v = v0 + u * 256;
        val[u] = make_uint4(0, 0, 0, 0);
        dst[u] = -1;
        if (v < dy_vecs) {
          int m = v / NVEC, vec = v - m * NVEC;
          int ti = fdiv(m, t.fd_thw), rem = m - ti * thw;
          int ty = fdiv(rem, t.fd_tw), tx = rem - ty * t.TW;
          int n = n0 + ti, oy = oy0 + ty, ox = ox0 + tx;
          int c = co0 + vec * VEC;
          dst[u] = m * P + vec * 16;
          if (ti < t.TI && n < d.N && oy < d.Ho && ox < d.Wo && c < d.Cout) {
            size_t off = (((size_t)n * d.Ho + oy) * d.Wo + ox) * (size_t)d.dy_pitch + c;
            val[u] = *(const uint4*)(dyg + off * SZ);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < WSTAGE_U; ++u)
        if (dst[u] >= 0) *(uint4*)(lds_dy + dst[u]) = val[u];
    }
    __syncthreads();
    for (int kb = 0; kb < ppw; kb += K) {
      if constexpr (SZ == 2) {
        // k-slot (khalf, q4, j) <-> pixel  wave*ppw + kb + 8*khalf + 4*q4 + j
        const int m0 = wave * ppw + kb + 8 * khalf + tr_pix;
        const int x0 = xtab[m0] + tr_coff, x1 = xtab[m0 + 4] + tr_coff;
        const char* pa = lds_dy + m0 * P + tr_coff;
        bf16x4 a0 = WG<T>::tr(pa), a1 = WG<T>::tr(pa + 4 * P);
        bf16x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int tp = 0; tp < NT; ++tp) {
          bf16x4 b0 = WG<T>::tr(lds_x + x0 + tapoff[tp]), b1 = WG<T>::tr(lds_x + x1 + tapoff[tp]);
          bf16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
          WG<T>::mma(a, b, acc[tp]);
        }
      } else {
        const int m0 = wave * ppw + kb + khalf;
        const float a = *(const float*)(lds_dy + m0 * P + l31 * 4);
        const int xo = xtab[m0] + l31 * 4;
#pragma unroll
        for (int tp = 0; tp < NT; ++tp) {
          const float b = *(const float*)(lds_x + xo + tapoff[tp]);
          WG<T>::mma(a, b, acc[tp]);
        }
      }
    }
  }
  // ---- combine the 4 waves' partial sums in LDS, then fp32 atomics to dW -------------------------
  __syncthreads();
  for (int i = tid; i < NT * 1024; i += 256) red[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int tp = 0; tp < NT; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;  // cout within block
          red[(tp * 32 + row) * 32 + l31] += acc[tp][r];
        }
    }
    __syncthreads();
  }
  if (t.use_ws) {
    // partial slab [g][block][NT*1024], coalesced; a second launch folds the G slabs into dW
    float* ws = (float*)d.workspace + ((size_t)blockIdx.x * gridDim.y + blk) * (NT * 1024);
    for (int i = tid; i < NT * 1024; i += 256) ws[i] = red[i];
  } else {
    for (int i = tid; i < NT * 1024; i += 256) {
      int ci = i & 31, row = (i >> 5) & 31, tp = i >> 10;
      int co = co0 + row, cin = ci0 + ci;
      if (co < d.Cout && cin < d.dw_cin) atomicAdd(&d.dw[((size_t)co * d.dw_cin + cin) * d.ntaps + tp], red[i]);
    }
  }
}

// dW[co][ci][tp] (+)= sum_g ws[g][blk][tp][row][ci]
template <int NT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const hrp_wgrad_desc d, int G, int pairs, int n_cib) {
  const int blk = blockIdx.y;
  const int cob = blk / n_cib, cib = blk - cob * n_cib;
  // block = 64 consecutive elements x 4 slab phases; lanes read 256 contiguous bytes of a slab, 8 loads
  // in flight per thread
  __shared__ float part[4][64];
  const float* ws = (const float*)d.workspace + (size_t)blk * (NT * 1024);
  const size_t gstride = (size_t)pairs * (NT * 1024);
  const int e = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  float s = 0.f;
  int g = ph;
  for (; g + 28 < G; g += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(g + 4 * u) * gstride + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; g < G; g += 4) s += ws[(size_t)g * gstride + i];
  part[ph][e] = s;
  __syncthreads();
  if (ph == 0) {
    s = part[0][e] + part[1][e] + part[2][e] + part[3][e];
    int ci = i & 31, row = (i >> 5) & 31, tp = i >> 10;
    int co = cob * 32 + row, cin = cib * 32 + ci;
    if (co < d.Cout && cin < d.dw_cin) {
      float* o = &d.dw[((size_t)co * d.dw_cin + cin) * d.ntaps + tp];
      *o = d.accumulate ? *o + s : s;
    }
  }
}

template <typename T, int NT>
static int wgrad_tiling(const hrp_wgrad_desc& d, WgradTiling& t) {
  constexpr int SZ = Elem<T>::SZ;
  constexpr int P = 32 * SZ + 16;
  int mindy = 1 << 30, maxdy = -(1 << 30), mindx = 1 << 30, maxdx = -(1 << 30);
  for (int i = 0; i < d.ntaps; ++i) {
    mindy = d.dy_t[i] < mindy ? d.dy_t[i] : mindy; maxdy = d.dy_t[i] > maxdy ? d.dy_t[i] : maxdy;
    mindx = d.dx_t[i] < mindx ? d.dx_t[i] : mindx; maxdx = d.dx_t[i] > maxdx ? d.dx_t[i] : maxdx;
  }
  t.mindy = mindy; t.mindx = mindx;
  const int budget = 72 * 1024;
  int lds = 0;
  for (int BM = 256; BM >= 64; BM >>= 1) {
    int TW = 1; while (TW < d.Wo && TW < 16) TW <<= 1;
    int TH = 1; while (TH < d.Ho && TH * TW < BM) TH <<= 1;
    int TI = BM / (TW * TH);
    if (TI > d.N) TI = d.N;
    t.TW = TW; t.TH = TH; t.BM = BM;
    t.IHt = (TH - 1) * d.in_stride + (maxdy - mindy) + 1;
    t.IWt = (TW - 1) * d.in_stride + (maxdx - mindx) + 1;
    {
      int maxti = (budget - BM * P - BM * 4) / (t.IHt * t.IWt * P);
      if (maxti < 1) maxti = 1;
      if (TI > maxti) TI = maxti;
    }
    t.TI = TI;
    t.in_pix = TI * t.IHt * t.IWt;
    t.lds_dy_off = round_up(t.in_pix * P, 16);
    t.lds_tab_off = t.lds_dy_off + BM * P;
    int main_bytes = t.lds_tab_off + BM * 4;
    t.lds_red_off = 0;  // the reduction buffer reuses the tiles
    int red_bytes = NT * 1024 * 4;
    lds = main_bytes > red_bytes ? main_bytes : red_bytes;
    long pixels = (long)d.N * d.Ho * d.Wo;
    if (lds <= budget && (BM == 64 || pixels >= BM)) break;
    if (BM == 64 && lds > 160 * 1024) {
      set_error("wgrad: tile does not fit LDS");
      return HRP_ERR_ARG;
    }
  }
  t.lds_bytes = lds;
  t.tiles_x = cdiv(d.Wo, t.TW); t.tiles_y = cdiv(d.Ho, t.TH); t.tiles_n = cdiv(d.N, t.TI);
  t.ntiles = t.tiles_x * t.tiles_y * t.tiles_n;
  t.n_cob = cdiv(d.Cout, 32); t.n_cib = cdiv(d.Cin, 32);
  int pairs = t.n_cob * t.n_cib;
  // workgroups per (cout, cin) block: ~2 per CU over the whole launch; each walks ntiles / G pixel tiles
  int G = 256 / pairs;
  if (G < 1) G = 1;
  if (G > t.ntiles) G = t.ntiles;
  t.G = G;
  t.fd_ihw = make_fastdiv(t.IHt * t.IWt); t.fd_iwt = make_fastdiv(t.IWt);
  t.fd_thw = make_fastdiv(t.TH * t.TW); t.fd_tw = make_fastdiv(t.TW);
  t.fd_tx = make_fastdiv(t.tiles_x); t.fd_ty = make_fastdiv(t.tiles_y); t.fd_cib = make_fastdiv(t.n_cib);
  return HRP_OK;
}

template <typename T, int NT>
static int64_t wgrad_ws_bytes(const hrp_wgrad_desc& d) {
  WgradTiling t{};
  if (wgrad_tiling<T, NT>(d, t) != HRP_OK) return 0;
  return (int64_t)t.G * t.n_cob * t.n_cib * NT * 1024 * 4;
}

template <typename T, int NT>
static int launch_wgrad(const hrp_wgrad_desc& d, hipStream_t s) {
  WgradTiling t{};
  int rc = wgrad_tiling<T, NT>(d, t);
  if (rc != HRP_OK) return rc;
  const int pairs = t.n_cob * t.n_cib;
  const int64_t need = (int64_t)t.G * pairs * NT * 1024 * 4;
  t.use_ws = (d.workspace && d.workspace_bytes >= need) ? 1 : 0;
  if (!t.use_ws && !d.accumulate)
    (void)hipMemsetAsync(d.dw, 0, sizeof(float) * (size_t)d.Cout * d.dw_cin * d.ntaps, s);
  auto kern = conv_wgrad_kernel<T, NT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(t.G, pairs), dim3(256), t.lds_bytes, s, d, t);
  rc = check_launch("conv_wgrad_kernel");
  if (rc != HRP_OK || !t.use_ws) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel<NT>, dim3(NT * 1024 / 64, pairs), dim3(256), 0, s, d, t.G, pairs, t.n_cib);
  return check_launch("wgrad_reduce_kernel");
}

}  // namespace hrp

extern "C" int hrp_conv2d_bwd_weight(const hrp_wgrad_desc* d, void* stream) {
  using namespace hrp;
  HRP_REQUIRE(d && d->x && d->dy && d->dw, "wgrad: null pointer");
  HRP_REQUIRE(d->dtype == HRP_F32 || d->dtype == HRP_BF16, "wgrad: dtype");
  const int vec = d->dtype == HRP_F32 ? 4 : 8;
  HRP_REQUIRE(d->Cin % vec == 0 && d->x_pitch % vec == 0 && (uintptr_t)d->x % 16 == 0, "wgrad: x channels/pitch/alignment");
  // dy may have any Cout as long as whole 16-byte vectors can be read (garbage lanes are masked at the store)
  HRP_REQUIRE(d->dy_pitch % vec == 0 && d->dy_pitch >= (d->Cout + vec - 1) / vec * vec && (uintptr_t)d->dy % 16 == 0,
              "wgrad: dy pitch must be a multiple of %d and cover Cout rounded up (Cout=%d pitch=%d)", vec, d->Cout, d->dy_pitch);
  HRP_REQUIRE(d->ntaps == 1 || d->ntaps == 4 || d->ntaps == 9, "wgrad: ntaps=%d unsupported", d->ntaps);
  HRP_REQUIRE(d->dw_cin <= d->Cin, "wgrad: dw_cin > Cin");
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == HRP_F32) {
    if (d->ntaps == 1) return launch_wgrad<float, 1>(*d, s);
    if (d->ntaps == 4) return launch_wgrad<float, 4>(*d, s);
    return launch_wgrad<float, 9>(*d, s);
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
  if (d->ntaps == 1) return wgrad_ws_bytes<bf16_t, 1>(*d);
  if (d->ntaps == 4) return wgrad_ws_bytes<bf16_t, 4>(*d);
  return wgrad_ws_bytes<bf16_t, 9>(*d);
}
