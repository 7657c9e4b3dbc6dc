// The iterative regressors of the full network (reference lib/models/full_net.py:318-331 joint angles, :365-378 rotation):
//     p_{i+1} = p_i + dec(drop(fc2(drop(fc1(cat(xf, p_i))))))          n_iter = 4 times, NO non-linearity between the layers
// Round 1-5 ran every nn.Linear / cat / dropout / residual of that loop as a launch of its own (169 launches per training
// step at the join of both trunks, where nothing overlaps them).  Here the loop is a chain of ONE kernel:
//
//   hoist      A = xf W1[:, :F]^T + b1                      once per forward (the reference recomputes it per iteration)
//   step i     p_i   = p_{i-1} + b3 + d2_{i-1} W3^T         every workgroup, from the previous step's output (fixed order)
//              d1_i  = m1_i o (A + p_i W1[:, F:]^T)         while the row operand is staged (rank-P update + dropout mask)
//              d2_i  = m2_i o (b2 + d1_i W2^T)              the 64 x 1024 x 1024 product on v_mfma_f32_16x16x4_f32, exact fp32
//   backward   the same kernel on the transposed problem:  g_i = g_{i+1} + gh1_i W1[:, F:],  gh2_i = m2_i o (g_{i+1} W3),
//              gh1_i = m1_i o (gh2_i W2),  gA += gh1_i;  the weight gradients are left to the END of the chain, where the n_iter
//              iterations are ONE product over n_iter * M rows (hrp_linear_wgrad_batch), and d xf = sum over the heads of gA W1[:, :F]
//              is one more launch of this kernel with two sources.
// Both heads (and any number of problems <= HRP_REG_MAX_PROBLEMS) share every launch (blockIdx.y).
//
// A step is two kernels behind one entry point: a row-parallel one for the state and the operand (one workgroup per sample: the state
// is a 1024-long dot product per (sample, p) - inside the product kernel every workgroup repeated it from 256 KB of L2 reads and the
// launch took 63 us), and the product kernel (32 rows x 16 columns per workgroup, 256 workgroups for the two heads at B = 64).
// Every sum has a fixed order (no atomics, no split across workgroups): bit-reproducible.
#include "hrp_common.h"

namespace hrp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RG_KC = 128;             // reduction indices per chunk
constexpr int RG_PA = RG_KC + 4;       // LDS row pitch in floats (16-byte aligned rows)
constexpr int RG_ROWS = 32, RG_COLS = 16;

struct RegArgs {
  hrp_regressor_step_desc d[HRP_REG_MAX_PROBLEMS];
};

__device__ __forceinline__ float4 ld4(const float* p, bool vec) {
  if (vec) return *(const float4*)p;
  return make_float4(p[0], p[1], p[2], p[3]);
}

// ---- steps 1 + 2 of hrp_regressor_step_desc: one workgroup per row m.  u[p] = u_prev + u_bias + sum_k z[k] zw[k][p] (a thread
// takes k = tid, tid + 256, ..; wave tree, then the four waves in order: fixed order), then the operand row
// a'[k] = a_mask[k] (a[k] + sum_p u[p] v[k][p]) into a_out.  PMAX: compile-time bound of P.
template <int PMAX>
__global__ __launch_bounds__(256) void regressor_prep_kernel(const RegArgs args) {
  const hrp_regressor_step_desc& d = args.d[blockIdx.y];
  const int m = blockIdx.x;
  if (m >= d.M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = d.P;
  // the state sum in float64: it is the LAST arithmetic in front of the network's pose / rotation outputs (p_n = p_{n-1} + b3 +
  // d2 W3^T), and 1 um of a key-point 0.13 m in front of the camera is 3e-3 px - an fp32 chain of 1024 terms moved the fp32
  // key-points of the fp64-floor test by 1e-3 px with its summation order alone.  8 values per thread: free.
  __shared__ double red[4][PMAX];
  __shared__ float u_s[PMAX];
  if (P > 0) {
    if (d.z) {
      double acc[PMAX];
#pragma unroll
      for (int p = 0; p < PMAX; ++p) acc[p] = 0.0;
      const float* zr = d.z + (size_t)m * d.z_pitch;
      for (int kb = 0; kb < d.z_len; kb += 1024) {          // four k per thread and trip, every load of the trip issued before the first use
        float zk[4], wv[4][PMAX];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = kb + tid + 256 * j;
          const bool ok = k < d.z_len;
          zk[j] = ok ? zr[k] : 0.f;
          const float* wr = d.zw + (size_t)k * d.zw_sk;
#pragma unroll
          for (int p = 0; p < PMAX; ++p) wv[j][p] = (ok && p < P) ? wr[(size_t)p * d.zw_sp] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < PMAX; ++p) acc[p] = fma((double)zk[j], (double)wv[j][p], acc[p]);
      }
#pragma unroll
      for (int p = 0; p < PMAX; ++p) {
        double t = acc[p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) red[wave][p] = t;
      }
      __syncthreads();
    }
    if (tid < P) {
      double v = (double)d.u_prev[(size_t)m * P + tid];
      if (d.z) v += ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
      if (d.u_bias) v += (double)d.u_bias[tid];
      u_s[tid] = (float)v;
      if (d.u_out) d.u_out[(size_t)m * P + tid] = (float)v;
    }
    __syncthreads();
  }
  if (!d.a_out || d.N <= 0) return;
  float u[PMAX];
#pragma unroll
  for (int p = 0; p < PMAX; ++p) u[p] = (p < P && d.v) ? u_s[p] : 0.f;
  const float* ar = d.a ? d.a + (size_t)m * d.a_pitch : nullptr;
  const float* mr = d.a_mask ? d.a_mask + (size_t)m * d.K : nullptr;
  float* orow = d.a_out + (size_t)m * d.K;
  for (int kb = 0; kb < d.K; kb += 1024) {
    float val[4], mk[4], vv[4][PMAX];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kb + tid + 256 * j;
      const bool ok = k < d.K;
      val[j] = (ok && ar) ? ar[k] : 0.f;
      mk[j] = (ok && mr) ? mr[k] : 1.f;
      const float* vr = d.v ? d.v + (size_t)k * d.v_sk : nullptr;
#pragma unroll
      for (int p = 0; p < PMAX; ++p) vv[j][p] = (ok && vr && p < P) ? vr[(size_t)p * d.v_sp] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kb + tid + 256 * j;
#pragma unroll
      for (int p = 0; p < PMAX; ++p) val[j] = fmaf(u[p], vv[j][p], val[j]);
      if (k < d.K) orow[k] = val[j] * mk[j];
    }
  }
}

// ---- step 3: out = out_mask o (bias + a' w^T (+ a2 w2^T)), a' = a_out when steps 1 / 2 transformed the operand, else a.
// Workgroup = 32 rows x 16 columns x the whole reduction range in chunks of 128 through LDS (the next chunk's global loads in
// flight under the MFMAs); wave = (row block w & 1, half w >> 1 of every chunk); the two halves are added through LDS at the end.
__global__ __launch_bounds__(256) void regressor_gemm_kernel(const RegArgs args) {
  const hrp_regressor_step_desc& d = args.d[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * RG_COLS, m0 = blockIdx.z * RG_ROWS;
  if (m0 >= d.M || n0 >= d.N) return;
  __shared__ __attribute__((aligned(16))) float As[RG_ROWS * RG_PA];
  __shared__ __attribute__((aligned(16))) float Ws[RG_COLS * RG_PA];
  const int rows = d.M - m0 < RG_ROWS ? d.M - m0 : RG_ROWS;
  const int c4 = tid & 31, r0 = tid >> 5;                  // staging: 4 reduction indices x rows r0, r0 + 8, r0 + 16, r0 + 24
  const bool wt = d.w_sn == 1 && d.w_sk != 1;              // weight stored [k][n] (data-gradient direction)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kq = lane >> 4;
  const int rb = wave & 1, kh = wave >> 1;
  const bool wave_on = rb * 16 < rows;
  const bool transformed = d.a_out && (d.a_mask || (d.v && d.P > 0));
  const int K = d.K;

  for (int src = 0; src < 2; ++src) {
    const float* a = src ? d.a2 : (transformed ? d.a_out : d.a);
    const float* w = src ? d.w2 : d.w;
    if (src && !a) break;
    const int a_pitch = src ? d.a2_pitch : (transformed ? K : d.a_pitch);
    const long long w_sk = (src && d.w2_sk) ? (long long)d.w2_sk : d.w_sk;
    const bool a_vec = ((uintptr_t)a % 16 == 0) && (a_pitch % 4 == 0);
    const bool w_vec = ((uintptr_t)w % 16 == 0) && ((wt ? w_sk : d.w_sn) % 4 == 0);

    float4 ra[4], rw[2];
    auto load_chunk = [&](int k0) {
      const int k = k0 + 4 * c4;
      const bool kin = k < K;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + 8 * j;
        ra[j] = (kin && r < rows) ? ld4(a + (size_t)(m0 + r) * a_pitch + k, a_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (!wt) {             // w[n][k]: rows n0 + (tid >> 5) + 8 j, four consecutive k
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + r0 + 8 * j;
          rw[j] = (kin && n < d.N) ? ld4(w + (size_t)n * d.w_sn + k, w_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {               // w[k][n]: reduction index k0 + (tid >> 2) + 64 j, four consecutive n
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int kk = k0 + (tid >> 2) + 64 * j, n = n0 + 4 * (tid & 3);
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kk < K) {
            const float* q = w + (size_t)kk * w_sk + n;
            if (n + 3 < d.N) t = ld4(q, w_vec);
            else {
              if (n < d.N) t.x = q[0];
              if (n + 1 < d.N) t.y = q[1];
              if (n + 2 < d.N) t.z = q[2];
            }
          }
          rw[j] = t;
        }
      }
    };
    auto store_chunk = [&]() {
#pragma unroll
      for (int j = 0; j < 4; ++j) *(float4*)(As + (r0 + 8 * j) * RG_PA + 4 * c4) = ra[j];
      if (!wt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *(float4*)(Ws + (r0 + 8 * j) * RG_PA + 4 * c4) = rw[j];
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int kl = (tid >> 2) + 64 * j, nl = 4 * (tid & 3);
          Ws[nl * RG_PA + kl] = rw[j].x;
          Ws[(nl + 1) * RG_PA + kl] = rw[j].y;
          Ws[(nl + 2) * RG_PA + kl] = rw[j].z;
          Ws[(nl + 3) * RG_PA + kl] = rw[j].w;
        }
      }
    };

    load_chunk(0);
    for (int k0 = 0; k0 < K; k0 += RG_KC) {
      store_chunk();
      __syncthreads();
      if (k0 + RG_KC < K) load_chunk(k0 + RG_KC);
      if (wave_on) {
        const float* ap = As + (rb * 16 + l15) * RG_PA + kh * (RG_KC / 2) + kq;
        const float* bp = Ws + l15 * RG_PA + kh * (RG_KC / 2) + kq;
#pragma unroll
        for (int ks = 0; ks < RG_KC / 8; ks += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks], bp[4 * ks], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks + 4], bp[4 * ks + 4], acc1, 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  // ---- the second half of every chunk joins the first through LDS, then the epilogue: lane = column n0 + (lane & 15),
  // rows 16 rb + 4 (lane >> 4) + i
  f32x4 acc = acc0 + acc1;
  float* xch = As;                                          // [2 row blocks][64 lanes][4]
  if (kh == 1) *(f32x4*)(xch + (rb * 64 + lane) * 4) = acc;
  __syncthreads();
  if (kh == 1 || !wave_on) return;
  acc += *(const f32x4*)(xch + (rb * 64 + lane) * 4);
  const int n = n0 + l15;
  if (n >= d.N) return;
  const float bv = d.bias ? d.bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rb * 16 + 4 * kq + i;
    if (r >= rows) continue;
    const size_t m = (size_t)(m0 + r);
    float val = acc[i] + bv;
    if (d.out_mask) val *= d.out_mask[m * d.N + n];
    if (d.out_sum) {
      float* q = d.out_sum + m * d.N + n;
      *q = d.out_sum_accumulate ? *q + val : val;
    }
    float* o = d.out + m * d.out_pitch + n;
    *o = d.out_accumulate ? *o + val : val;
  }
}

// ---- weight gradients of several linear layers in one launch: dw[n][k] (+)= sum_m dy[m][n] x[m][k], dbias[n] (+)= sum_m dy[m][n] ----
struct LinWgradArgs {
  hrp_linear_wgrad_desc d[HRP_LIN_WGRAD_MAX];
  int first[HRP_LIN_WGRAD_MAX + 1];      // first workgroup of every problem
  int n;
};

__global__ __launch_bounds__(256) void linear_wgrad_batch_kernel(const LinWgradArgs args) {
  __shared__ float Ds[64 * 65];   // dy tile [m][n]
  __shared__ float Xs[64 * 65];   // x tile  [m][k]
  int pi = 0;
  while (pi + 1 < args.n && (int)blockIdx.x >= args.first[pi + 1]) ++pi;
  const hrp_linear_wgrad_desc& d = args.d[pi];
  const int local = blockIdx.x - args.first[pi];
  const int kt = (d.K + 63) / 64;
  const int n0 = (local / kt) * 64, k0 = (local % kt) * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
  const int l31 = lane & 31, kh = lane >> 5;
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  const bool want_b = d.dbias && k0 == 0 && tid < 64;
  for (int m0 = 0; m0 < d.M; m0 += 64) {
    if (m0) __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
      const int m = i >> 6, c = i & 63;
      Ds[m * 65 + c] = (m0 + m < d.M && n0 + c < d.N) ? d.dy[(size_t)(m0 + m) * d.dy_pitch + n0 + c] : 0.f;
      Xs[m * 65 + c] = (m0 + m < d.M && k0 + c < d.K) ? d.x[(size_t)(m0 + m) * d.x_pitch + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int mm = 0; mm < 64; mm += 2) {
      const float av = Ds[(mm + kh) * 65 + wn + l31];
      const float bv = Xs[(mm + kh) * 65 + wk + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (want_b) {
      for (int m = 0; m < 64; ++m) bsum += Ds[m * 65 + tid];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = n0 + wn + (i & 3) + 8 * (i >> 2) + 4 * kh, k = k0 + wk + l31;
    if (n < d.N && k < d.K) {
      float* o = d.dw + (size_t)n * d.dw_ld + k;
      *o = d.accumulate ? *o + acc[i] : acc[i];
    }
  }
  if (want_b && n0 + tid < d.N) d.dbias[n0 + tid] = d.accumulate ? d.dbias[n0 + tid] + bsum : bsum;
}

// ---- every dropout mask of a forward in one launch ----------------------------------------------------------------------------
__device__ __forceinline__ void philox_round_r(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void dropout_masks_kernel(float* __restrict__ mask, size_t n, float keep,
                                                            const uint64_t* __restrict__ state, uint32_t salt) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;     // 4 consecutive elements
  if (q * 4 >= n) return;
  const uint64_t seed = state[0], step = state[1];
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32) ^ salt, (uint32_t)step, (uint32_t)(step >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round_r(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float scale = 1.0f / keep;
  float m[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j] = ((float)(c[j] >> 8) * (1.0f / 16777216.0f)) < keep ? scale : 0.f;
  if (q * 4 + 3 < n) *(float4*)(mask + q * 4) = make_float4(m[0], m[1], m[2], m[3]);
  else
    for (int j = 0; j < 4 && q * 4 + j < n; ++j) mask[q * 4 + j] = m[j];
}

}  // namespace hrp

using namespace hrp;

extern "C" int hrp_regressor_step(const hrp_regressor_step_desc* descs, int n, void* stream) {
  HRP_REQUIRE(descs && n >= 1 && n <= HRP_REG_MAX_PROBLEMS, "regressor_step: 1 .. %d problems", HRP_REG_MAX_PROBLEMS);
  RegArgs args;
  int gx = 0, gz = 1, pmax = 0, mmax = 0;
  bool prep = false;
  for (int i = 0; i < n; ++i) {
    const hrp_regressor_step_desc& d = descs[i];
    HRP_REQUIRE(d.M > 0 && d.N >= 0 && d.P >= 0 && d.P <= HRP_REG_MAX_P, "regressor_step[%d]: M %d, N %d, P %d", i, d.M, d.N, d.P);
    if (d.P > 0) {
      HRP_REQUIRE(d.u_prev, "regressor_step[%d]: a state of width %d needs u_prev", i, d.P);
      HRP_REQUIRE(!d.z || (d.zw && d.z_len > 0), "regressor_step[%d]: z needs zw and a length", i);
      prep = true;
    }
    if (d.N > 0) {
      HRP_REQUIRE(d.w && d.out && d.K > 0 && d.K % 4 == 0 && d.out_pitch >= d.N, "regressor_step[%d]: product needs w, out, K %% 4 == 0", i);
      HRP_REQUIRE(d.a || (d.v && d.P > 0), "regressor_step[%d]: no row operand", i);
      HRP_REQUIRE(d.w_sn == 1 || d.w_sk == 1, "regressor_step[%d]: the weight must be contiguous along n or along k", i);
      const bool transform = d.a_mask || (d.v && d.P > 0);
      HRP_REQUIRE(!transform || (d.a_out && (uintptr_t)d.a_out % 16 == 0),
                  "regressor_step[%d]: a masked / updated operand is staged through a_out (dense [M][K], 16-byte aligned)", i);
      HRP_REQUIRE(d.a || transform, "regressor_step[%d]: no row operand", i);
      HRP_REQUIRE(!d.a2 || d.w2, "regressor_step[%d]: a2 without w2", i);
      if (transform || d.a_out) prep = true;
      if ((d.N + RG_COLS - 1) / RG_COLS > gx) gx = (d.N + RG_COLS - 1) / RG_COLS;
    }
    if ((d.M + RG_ROWS - 1) / RG_ROWS > gz) gz = (d.M + RG_ROWS - 1) / RG_ROWS;
    if (d.P > pmax) pmax = d.P;
    if (d.M > mmax) mmax = d.M;
    args.d[i] = d;
  }
  hipStream_t s = (hipStream_t)stream;
  if (prep) {
    const dim3 grid(mmax, n), blk(256);
    if (pmax <= 8) hipLaunchKernelGGL((regressor_prep_kernel<8>), grid, blk, 0, s, args);
    else hipLaunchKernelGGL((regressor_prep_kernel<16>), grid, blk, 0, s, args);
  }
  if (gx > 0) hipLaunchKernelGGL(regressor_gemm_kernel, dim3(gx, n, gz), dim3(256), 0, s, args);
  return check_launch("regressor_step");
}

extern "C" int hrp_linear_wgrad_batch(const hrp_linear_wgrad_desc* descs, int n, void* stream) {
  HRP_REQUIRE(descs && n >= 1 && n <= HRP_LIN_WGRAD_MAX, "linear_wgrad_batch: 1 .. %d problems", HRP_LIN_WGRAD_MAX);
  LinWgradArgs args;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const hrp_linear_wgrad_desc& d = descs[i];
    HRP_REQUIRE(d.x && d.dy && d.dw && d.M > 0 && d.K > 0 && d.N > 0 && d.x_pitch >= d.K && d.dy_pitch >= d.N && d.dw_ld >= d.K,
                "linear_wgrad_batch[%d]: bad arguments", i);
    args.d[i] = d;
    args.first[i] = total;
    total += cdiv(d.K, 64) * cdiv(d.N, 64);
  }
  args.first[n] = total;
  args.n = n;
  hipLaunchKernelGGL(linear_wgrad_batch_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, args);
  return check_launch("linear_wgrad_batch");
}

extern "C" int hrp_dropout_masks(float* masks, int64_t n, float keep, const uint64_t* state_dev, uint32_t salt, void* stream) {
  HRP_REQUIRE(masks && n > 0 && state_dev && (uintptr_t)masks % 16 == 0, "dropout_masks: bad args");
  HRP_REQUIRE(keep > 0.f && keep <= 1.f, "dropout_masks: keep probability %f", (double)keep);
  const size_t quads = ((size_t)n + 3) / 4;
  hipLaunchKernelGGL(dropout_masks_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, masks, (size_t)n, keep,
                     state_dev, salt);
  return check_launch("dropout_masks");
}
