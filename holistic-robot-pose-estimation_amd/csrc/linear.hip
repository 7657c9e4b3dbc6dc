// nn.Linear of the regression heads (reference lib/models/full_net.py:95-100, 129-134, 159-165; the iterative
// regressors of :318-331, 365-378 call them 24 times per forward) as skinny fp32 GEMMs: M = batch rows (64), the
// weight [N][K] is read ONCE per launch straight from the PyTorch-shaped parameter (no packing), exact fp32 on
// v_mfma_f32_32x32x2_f32.
//
//   forward        y[M,N]  = x[M,K] W^T + b (+ res)          reduce over K   (hrp_linear_fwd)
//   data gradient  dx[M,K] (+)= dy[M,N] W                    reduce over N   (hrp_linear_bwd_data)
//   weight grad    dW[N,K] (+)= dy^T x,  db[N] (+)= sum dy   reduce over M   (hrp_linear_bwd_weight)
//
// forward / data gradient: one workgroup = 64 rows x 32 output columns x 128 reduction indices (tiles staged in LDS
// with coalesced loads, each of the 4 waves multiplies a quarter of the reduction range, partial tiles are summed
// through LDS); the reduction dimension is split over workgroups so that >= 256 of them stream the weight, partial
// sums leave as fp32 atomics into a zeroed output (bias / residual ride on the first split).
// weight gradient: one workgroup = a 64 x 64 block of dW (4 waves x 32 x 32), reduction over the M rows, read-modify-
// write of dW with 128-byte coalesced rows; the workgroups of the first K block also produce the bias gradient.
#include "hrp_common.h"

namespace hrp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LBK = 128;            // reduction indices per workgroup
constexpr int LPA = LBK + 1;        // padded row of the row-operand tile (lanes = rows: conflict-free 4-byte reads)

// TRANS = false: out[m][c] += sum_r a[m][r] * w[c][r]   (forward: c = output feature, r = input feature)
// TRANS = true : out[m][c] += sum_r a[m][r] * w[r][c]   (data gradient: c = input feature, r = output feature)
template <bool TRANS>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ a, int a_pitch, const float* __restrict__ w, int w_ld,
                                                     const float* __restrict__ bias, const float* __restrict__ res, int res_pitch,
                                                     float* __restrict__ out, int out_pitch, int M, int R, int Cn, int use_atomics,
                                                     float* __restrict__ ws_part) {
  __shared__ float As[4 * 64 * 33 > 64 * LPA ? 4 * 64 * 33 : 64 * LPA];   // row-operand tile, later the 4 partial output tiles
  __shared__ float Bs[TRANS ? LBK * 33 : 32 * LPA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * LBK, m0 = blockIdx.z * 64;
  // ---- stage the tiles (zero beyond the edges) -----------------------------------------------------------
  for (int i = tid; i < 64 * LBK; i += 256) {
    const int m = i / LBK, r = i - m * LBK;
    As[m * LPA + r] = (m0 + m < M && r0 + r < R) ? a[(size_t)(m0 + m) * a_pitch + r0 + r] : 0.f;
  }
  if (!TRANS) {
    for (int i = tid; i < 32 * LBK; i += 256) {
      const int c = i / LBK, r = i - c * LBK;
      Bs[c * LPA + r] = (c0 + c < Cn && r0 + r < R) ? w[(size_t)(c0 + c) * w_ld + r0 + r] : 0.f;
    }
  } else {
    for (int i = tid; i < LBK * 32; i += 256) {
      const int r = i >> 5, c = i & 31;
      Bs[r * 33 + c] = (c0 + c < Cn && r0 + r < R) ? w[(size_t)(r0 + r) * w_ld + c0 + c] : 0.f;
    }
  }
  __syncthreads();
  // ---- each wave: a quarter of the reduction range, two 32-row tiles -------------------------------------------
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
  const int l31 = lane & 31, kh = lane >> 5;
  const int rb = wave * (LBK / 4);
#pragma unroll 4
  for (int kk = 0; kk < LBK / 4; kk += 2) {
    const int r = rb + kk + kh;
    const float a0 = As[l31 * LPA + r], a1 = As[(32 + l31) * LPA + r];
    const float b = TRANS ? Bs[r * 33 + l31] : Bs[l31 * LPA + r];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
  }
  __syncthreads();
  // ---- sum the 4 partial tiles through LDS (reusing the row-operand tile: 4 x 64 x 33 floats fit) ----------------
  float* part = As;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
    part[(wave * 64 + row) * 33 + l31] = acc0[i];
    part[(wave * 64 + 32 + row) * 33 + l31] = acc1[i];
  }
  __syncthreads();
  if (ws_part) {
    // Deterministic reduction over the splits: every workgroup parks its partial tile in the workspace, linear_fold_kernel adds
    // them up in split order.  (fp32 atomics into the output - the path without a workspace - depend on arrival order; a
    // last-ticket reduction inside this kernel needs device-scope fences, which write back / invalidate the L2 of an XCD:
    // measured 63 us against 18.)
    const int Mp = gridDim.z * 64, Cp = gridDim.x * 32;
    for (int i = tid; i < 64 * 32; i += 256) {
      const int m = i >> 5, c = i & 31;
      ws_part[((size_t)blockIdx.y * Mp + m0 + m) * Cp + c0 + c] =
          part[m * 33 + c] + part[(64 + m) * 33 + c] + part[(128 + m) * 33 + c] + part[(192 + m) * 33 + c];
    }
    return;
  }
  for (int i = tid; i < 64 * 32; i += 256) {
    const int m = i >> 5, c = i & 31;
    if (m0 + m >= M || c0 + c >= Cn) continue;
    float v = part[m * 33 + c] + part[(64 + m) * 33 + c] + part[(128 + m) * 33 + c] + part[(192 + m) * 33 + c];
    if (blockIdx.y == 0) {
      if (bias) v += bias[c0 + c];
      if (res) v += res[(size_t)(m0 + m) * res_pitch + c0 + c];
    }
    float* o = out + (size_t)(m0 + m) * out_pitch + c0 + c;
    if (use_atomics) atomicAdd(o, v);
    else *o = v;
  }
}

// dW[n][k] (+)= sum_m dy[m][n] * x[m][k];  db[n] (+)= sum_m dy[m][n]
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ x, int x_pitch, const float* __restrict__ dy, int dy_pitch,
                                                           float* __restrict__ dw, float* __restrict__ db, int M, int K, int N, int accumulate) {
  __shared__ float Ds[64 * 65];   // dy tile [m][n]
  __shared__ float Xs[64 * 65];   // x tile  [m][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
  const int l31 = lane & 31, kh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  for (int m0 = 0; m0 < M; m0 += 64) {
    if (m0) __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
      const int m = i >> 6, c = i & 63;
      Ds[m * 65 + c] = (m0 + m < M && n0 + c < N) ? dy[(size_t)(m0 + m) * dy_pitch + n0 + c] : 0.f;
      Xs[m * 65 + c] = (m0 + m < M && k0 + c < K) ? x[(size_t)(m0 + m) * x_pitch + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int mm = 0; mm < 64; mm += 2) {
      const float av = Ds[(mm + kh) * 65 + wn + l31];
      const float bv = Xs[(mm + kh) * 65 + wk + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (db && blockIdx.x == 0 && tid < 64) {
      for (int m = 0; m < 64; ++m) bsum += Ds[m * 65 + tid];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = n0 + wn + (i & 3) + 8 * (i >> 2) + 4 * kh, k = k0 + wk + l31;
    if (n < N && k < K) {
      float* o = dw + (size_t)n * K + k;
      *o = accumulate ? *o + acc[i] : acc[i];
    }
  }
  if (db && blockIdx.x == 0 && tid < 64 && n0 + tid < N) db[n0 + tid] = accumulate ? db[n0 + tid] + bsum : bsum;
}

__global__ void zero_rows_kernel(float* __restrict__ p, int pitch, int rows, int cols) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols;
  p[(size_t)r * pitch + (i - r * cols)] = 0.f;
}

// workspace of the deterministic reduction: [rs][mb * 64][ct * 32] partial sums (every element is written by the main launch
// before this one reads it: no initialisation needed)
// out[m][c] (+)= sum over the splits y (in order) of part[y][m][c] + bias[c] + res[m][c]
__global__ __launch_bounds__(256) void linear_fold_kernel(const float* __restrict__ part, int rs, int Mp, int Cp, const float* __restrict__ bias,
                                                          const float* __restrict__ res, int res_pitch, float* __restrict__ out, int out_pitch,
                                                          int M, int Cn, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * Cn) return;
  const int m = i / Cn, c = i - m * Cn;
  const float* pp = part + (size_t)m * Cp + c;
  const size_t ystride = (size_t)Mp * Cp;
  float v = 0.f;
  for (int y0 = 0; y0 < rs; y0 += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = y0 + u < rs ? pp[(size_t)(y0 + u) * ystride] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  if (bias) v += bias[c];
  if (res) v += res[(size_t)m * res_pitch + c];
  float* o = out + (size_t)m * out_pitch + c;
  *o = accumulate ? *o + v : v;
}

// workspace of the deterministic reduction: [rs][mb * 64][ct * 32] partial sums (no initialisation needed)
static inline size_t linear_ws_bytes(int M, int R, int Cn) {
  return (size_t)cdiv(R, LBK) * cdiv(M, 64) * 64 * cdiv(Cn, 32) * 32 * 4;
}

static int linear_launch(bool trans, const float* a, int a_pitch, const float* w, int w_ld, const float* bias, const float* res,
                         int res_pitch, float* out, int out_pitch, int M, int R, int Cn, int accumulate, void* ws, int64_t ws_bytes,
                         hipStream_t s) {
  const int ct = cdiv(Cn, 32), rs = cdiv(R, LBK), mb = cdiv(M, 64);
  const dim3 grid(ct, rs, mb);
  if (ws && rs > 1) {
    HRP_REQUIRE((uintptr_t)ws % 16 == 0 && (size_t)ws_bytes >= linear_ws_bytes(M, R, Cn), "linear: workspace too small (%lld < %zu bytes)",
                (long long)ws_bytes, linear_ws_bytes(M, R, Cn));
    float* part = (float*)ws;
    if (trans) hipLaunchKernelGGL(linear_kernel<true>, grid, dim3(256), 0, s, a, a_pitch, w, w_ld, nullptr, nullptr, 0, out, out_pitch, M, R, Cn, 0, part);
    else hipLaunchKernelGGL(linear_kernel<false>, grid, dim3(256), 0, s, a, a_pitch, w, w_ld, nullptr, nullptr, 0, out, out_pitch, M, R, Cn, 0, part);
    hipLaunchKernelGGL(linear_fold_kernel, dim3(cdiv(M * Cn, 256)), dim3(256), 0, s, part, rs, mb * 64, ct * 32, bias, res, res_pitch, out, out_pitch,
                       M, Cn, accumulate);
    return check_launch("linear");
  }
  const int atomics = (rs > 1 || accumulate) ? 1 : 0;
  if (atomics && !accumulate) {     // partial sums of the reduction splits meet in a zeroed output
    if (out_pitch == Cn) zero_async(out, sizeof(float) * (size_t)M * Cn, s);
    else hipLaunchKernelGGL(zero_rows_kernel, dim3(cdiv(M * Cn, 256)), dim3(256), 0, s, out, out_pitch, M, Cn);
  }
  if (trans) hipLaunchKernelGGL(linear_kernel<true>, grid, dim3(256), 0, s, a, a_pitch, w, w_ld, bias, res, res_pitch, out, out_pitch, M, R, Cn, atomics, nullptr);
  else hipLaunchKernelGGL(linear_kernel<false>, grid, dim3(256), 0, s, a, a_pitch, w, w_ld, bias, res, res_pitch, out, out_pitch, M, R, Cn, atomics, nullptr);
  return check_launch("linear");
}

}  // namespace hrp

using namespace hrp;

extern "C" int64_t hrp_linear_workspace_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  const size_t f = linear_ws_bytes(M, K, N), b = linear_ws_bytes(M, N, K);      // forward reduces over K, the data gradient over N
  return (int64_t)(f > b ? f : b);
}

extern "C" int hrp_linear_fwd(const float* x, int x_pitch, const float* w, const float* bias, const float* res, int res_pitch,
                              float* y, int y_pitch, int M, int K, int N, void* workspace, int64_t workspace_bytes, void* stream) {
  HRP_REQUIRE(x && w && y && M > 0 && K > 0 && N > 0 && x_pitch >= K && y_pitch >= N, "linear_fwd: bad arguments");
  HRP_REQUIRE(!res || (res_pitch >= N && res != y), "linear_fwd: residual");
  return linear_launch(false, x, x_pitch, w, K, bias, res, res_pitch, y, y_pitch, M, K, N, 0, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int hrp_linear_bwd_data(const float* dy, int dy_pitch, const float* w, float* dx, int dx_pitch, int M, int K, int N,
                                   int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  HRP_REQUIRE(dy && w && dx && M > 0 && K > 0 && N > 0 && dy_pitch >= N && dx_pitch >= K, "linear_bwd_data: bad arguments");
  return linear_launch(true, dy, dy_pitch, w, K, nullptr, nullptr, 0, dx, dx_pitch, M, N, K, accumulate, workspace, workspace_bytes,
                       (hipStream_t)stream);
}

extern "C" int hrp_linear_bwd_weight(const float* x, int x_pitch, const float* dy, int dy_pitch, float* dw, float* dbias, int M, int K,
                                     int N, int accumulate, void* stream) {
  HRP_REQUIRE(x && dy && dw && M > 0 && K > 0 && N > 0 && x_pitch >= K && dy_pitch >= N, "linear_bwd_weight: bad arguments");
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3(cdiv(K, 64), cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, x, x_pitch, dy, dy_pitch, dw,
                     dbias, M, K, N, accumulate);
  return check_launch("linear_wgrad");
}
