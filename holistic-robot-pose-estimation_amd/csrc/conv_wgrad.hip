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
// Operand gather: the reduction index is the pixel, but NHWC keeps channels contiguous, so the 8
// k-values a lane needs for the bf16 MFMA are 8 different pixels.  v1 gathers them with 16-bit LDS
// reads (fp32 needs one 32-bit read per operand and is unaffected).
#include "hrp_common.h"

namespace hrp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short8 __attribute__((ext_vector_type(8)));

struct WgradTiling {
  int TH, TW, TI, BM;
  int IHt, IWt, mindy, mindx;
  int tiles_x, tiles_y, tiles_n, ntiles;
  int n_cob, n_cib, G;
  int in_pix;
  int lds_dy_off, lds_tab_off, lds_red_off;
};

template <typename T>
struct WG;
template <>
struct WG<bf16_t> {
  static constexpr int K = 16, KH = 8;
  using Frag = bf16x8;
  __device__ static __forceinline__ Frag gather(const char* base, const int* off) {
    short8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = *(const short*)(base + off[j]);
    return __builtin_bit_cast(Frag, r);
  }
  __device__ static __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct WG<float> {
  static constexpr int K = 2, KH = 1;
  using Frag = float;
  __device__ static __forceinline__ Frag gather(const char* base, const int* off) { return *(const float*)(base + off[0]); }
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
  constexpr int K = WG<T>::K, KH = WG<T>::KH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_x = smem;
  char* lds_dy = smem + t.lds_dy_off;
  int* xtab = (int*)(smem + t.lds_tab_off);
  float* red = (float*)(smem + t.lds_red_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int blk = blockIdx.y;
  const int cob = blk / t.n_cib, cib = blk % t.n_cib;
  const int co0 = cob * 32, ci0 = cib * 32;
  const int IS = d.in_stride;
  const int thw = t.TH * t.TW, ihw = t.IHt * t.IWt;

  // pixel -> X-tile byte offset table
  for (int m = tid; m < t.BM; m += 256) {
    int ti = m / thw, rem = m - ti * thw;
    int ty = rem / t.TW, tx = rem - ty * t.TW;
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

  for (int tile = blockIdx.x; tile < t.ntiles; tile += t.G) {
    int tt = tile;
    const int tx_i = tt % t.tiles_x; tt /= t.tiles_x;
    const int ty_i = tt % t.tiles_y;
    const int tn_i = tt / t.tiles_y;
    const int n0 = tn_i * t.TI, oy0 = ty_i * t.TH, ox0 = tx_i * t.TW;
    const int iy0 = oy0 * IS + t.mindy, ix0 = ox0 * IS + t.mindx;
    __syncthreads();
    for (int v = tid; v < t.in_pix * NVEC; v += 256) {
      int pix = v / NVEC, vec = v - pix * NVEC;
      int ti = pix / ihw, rem = pix - ti * ihw;
      int iy = rem / t.IWt, ix = rem - iy * t.IWt;
      int n = n0 + ti, gy = iy0 + iy, gx = ix0 + ix;
      int c = ci0 + vec * VEC;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (n < d.N && gy >= 0 && gy < d.H && gx >= 0 && gx < d.W && c < d.Cin) {
        size_t off = (((size_t)n * d.H + gy) * d.W + gx) * (size_t)d.x_pitch + c;
        val = *(const uint4*)(xg + off * SZ);
      }
      *(uint4*)(lds_x + pix * P + vec * 16) = val;
    }
    for (int v = tid; v < t.BM * NVEC; v += 256) {
      int m = v / NVEC, vec = v - m * NVEC;
      int ti = m / thw, rem = m - ti * thw;
      int ty = rem / t.TW, tx = rem - ty * t.TW;
      int n = n0 + ti, oy = oy0 + ty, ox = ox0 + tx;
      int c = co0 + vec * VEC;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (ti < t.TI && n < d.N && oy < d.Ho && ox < d.Wo && c < d.Cout) {
        size_t off = (((size_t)n * d.Ho + oy) * d.Wo + ox) * (size_t)d.dy_pitch + c;
        val = *(const uint4*)(dyg + off * SZ);
      }
      *(uint4*)(lds_dy + m * P + vec * 16) = val;
    }
    __syncthreads();
    for (int kb = 0; kb < ppw; kb += K) {
      const int m0 = wave * ppw + kb + khalf * KH;
      int offa[KH], offx[KH];
#pragma unroll
      for (int j = 0; j < KH; ++j) {
        offa[j] = (m0 + j) * P + l31 * SZ;
        offx[j] = xtab[m0 + j] + l31 * SZ;
      }
      typename WG<T>::Frag a = WG<T>::gather(lds_dy, offa);
#pragma unroll
      for (int tp = 0; tp < NT; ++tp) {
        const int tapoff = ((d.dy_t[tp] - t.mindy) * t.IWt + (d.dx_t[tp] - t.mindx)) * P;
        typename WG<T>::Frag b = WG<T>::gather(lds_x + tapoff, offx);
        WG<T>::mma(a, b, acc[tp]);
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
  for (int i = tid; i < NT * 1024; i += 256) {
    int ci = i & 31, row = (i >> 5) & 31, tp = i >> 10;
    int co = co0 + row, cin = ci0 + ci;
    if (co < d.Cout && cin < d.dw_cin) atomicAdd(&d.dw[((size_t)co * d.dw_cin + cin) * d.ntaps + tp], red[i]);
  }
}

template <typename T, int NT>
static int launch_wgrad(const hrp_wgrad_desc& d, hipStream_t s) {
  constexpr int SZ = Elem<T>::SZ;
  constexpr int P = 32 * SZ + 16;
  WgradTiling t{};
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
  t.tiles_x = cdiv(d.Wo, t.TW); t.tiles_y = cdiv(d.Ho, t.TH); t.tiles_n = cdiv(d.N, t.TI);
  t.ntiles = t.tiles_x * t.tiles_y * t.tiles_n;
  t.n_cob = cdiv(d.Cout, 32); t.n_cib = cdiv(d.Cin, 32);
  int pairs = t.n_cob * t.n_cib;
  int G = 2048 / pairs;
  if (G < 1) G = 1;
  if (G > t.ntiles) G = t.ntiles;
  t.G = G;
  if (!d.accumulate) (void)hipMemsetAsync(d.dw, 0, sizeof(float) * (size_t)d.Cout * d.dw_cin * d.ntaps, s);
  auto kern = conv_wgrad_kernel<T, NT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(G, pairs), dim3(256), lds, s, d, t);
  return check_launch("conv_wgrad_kernel");
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
