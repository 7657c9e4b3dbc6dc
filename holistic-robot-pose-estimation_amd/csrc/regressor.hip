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
// Workgroup = 64 rows x 16 output columns x the whole reduction range: wave w owns rows 16 w .. 16 w + 15 (one 16 x 16 accumulator
// pair, even / odd k-steps), the reduction runs in chunks of 128 through LDS with the next chunk's global loads in flight under the
// MFMAs.  Every sum has a fixed order (k ascending inside a lane, no atomics, no split across workgroups): bit-reproducible.
#include "hrp_common.h"

namespace hrp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RG_KC = 128;             // reduction indices per chunk
constexpr int RG_PA = RG_KC + 4;       // LDS row pitch in floats (16-byte aligned rows)
constexpr int RG_ROWS = 64, RG_COLS = 16;
constexpr int RG_UP = 16;              // pitch of the state tile U[64][16]
constexpr int RG_TILE_FLOATS = RG_ROWS * RG_PA + RG_COLS * RG_PA;

struct RegArgs {
  hrp_regressor_step_desc d[HRP_REG_MAX_PROBLEMS];
};

__device__ __forceinline__ float4 ld4(const float* p, bool vec) {
  if (vec) return *(const float4*)p;
  return make_float4(p[0], p[1], p[2], p[3]);
}

// PMAX: compile-time bound of the state width P of every problem of the launch (0: no state anywhere)
template <int PMAX>
__global__ __launch_bounds__(256) void regressor_step_kernel(const RegArgs args) {
  const hrp_regressor_step_desc& d = args.d[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * RG_COLS, m0 = blockIdx.z * RG_ROWS;
  if (m0 >= d.M) return;
  if (d.N > 0 ? n0 >= d.N : blockIdx.x > 0) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* U = smem;                              // [64][RG_UP] state of the workgroup's rows
  float* As = smem + RG_ROWS * RG_UP;           // [64][RG_PA] row operand chunk
  float* Ws = As + RG_ROWS * RG_PA;             // [16][RG_PA] weight chunk
  float* ZW = As;                               // [P][z_len] during the state update (aliases the tiles)
  const int P = PMAX ? d.P : 0;
  const int rows = d.M - m0 < RG_ROWS ? d.M - m0 : RG_ROWS;

  // ---- 1. state: u = u_prev + u_bias + z zw -----------------------------------------------------------------------------
  if (PMAX && P > 0) {
    const int m = tid >> 2, pq = tid & 3;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (d.z) {
      const int zl = d.z_len;
      if (d.zw_sp == 1) {        // zw[k][p] rows of P contiguous values
        for (int i = tid; i < P * zl; i += 256) {
          const int k = i / P, p = i - k * P;
          ZW[p * zl + k] = d.zw[(size_t)k * d.zw_sk + p];
        }
      } else {                   // zw[p][k]: contiguous along k
        for (int i = tid; i < P * zl; i += 256) {
          const int p = i / zl, k = i - p * zl;
          ZW[i] = d.zw[(size_t)k * d.zw_sk + (size_t)p * d.zw_sp];
        }
      }
      __syncthreads();
      if (m < rows) {
        const float* zr = d.z + (size_t)(m0 + m) * d.z_pitch;
        for (int k4 = 0; k4 < zl; k4 += 16) {
          float4 zq[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) zq[u] = *(const float4*)(zr + k4 + 4 * u);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int p = pq + 4 * j;
            if (p < P) {
              const float* wr = ZW + p * zl + k4;
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float4 w4 = *(const float4*)(wr + 4 * u);
                acc[j] = fmaf(zq[u].x, w4.x, acc[j]);
                acc[j] = fmaf(zq[u].y, w4.y, acc[j]);
                acc[j] = fmaf(zq[u].z, w4.z, acc[j]);
                acc[j] = fmaf(zq[u].w, w4.w, acc[j]);
              }
            }
          }
        }
      }
      __syncthreads();           // ZW is dead: the tiles may be staged over it
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = pq + 4 * j;
      float v = 0.f;
      if (p < P && m < rows) {
        v = d.u_prev[(size_t)(m0 + m) * P + p] + acc[j];
        if (d.u_bias) v += d.u_bias[p];
        if (d.u_out && blockIdx.x == 0) d.u_out[(size_t)(m0 + m) * P + p] = v;
      }
      U[m * RG_UP + p] = v;
    }
    __syncthreads();
  }
  if (d.N <= 0) return;

  // ---- 2. / 3. the product, reduction in chunks of RG_KC ------------------------------------------------------------------
  const int c4 = tid & 31, r0 = tid >> 5;                  // row-operand staging: 4 reduction indices x rows r0, r0 + 8, ..
  const bool wt = d.w_sn == 1 && d.w_sk != 1;              // weight stored [k][n] (data-gradient direction)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kq = lane >> 4;
  const bool wave_on = wave * 16 < rows;

  for (int src = 0; src < 2; ++src) {
    const float* a = src ? d.a2 : d.a;
    const float* w = src ? d.w2 : d.w;
    if (src && !a) break;
    const int a_pitch = src ? d.a2_pitch : d.a_pitch;
    const float* amask = src ? nullptr : d.a_mask;
    const float* v = (PMAX && !src && P > 0) ? d.v : nullptr;
    float* a_out = (!src && blockIdx.x == 0) ? d.a_out : nullptr;
    const bool a_vec = a && ((uintptr_t)a % 16 == 0) && (a_pitch % 4 == 0);
    const long long w_sk = (src && d.w2_sk) ? (long long)d.w2_sk : d.w_sk;
    const bool w_vec = ((uintptr_t)w % 16 == 0) && ((wt ? w_sk : d.w_sn) % 4 == 0);
    const int K = d.K;

    float4 ra[8], rm[8], rw[2];
    float rv[4][PMAX ? PMAX : 1];
    auto load_chunk = [&](int k0) {
      const int k = k0 + 4 * c4;
      const bool kin = k < K;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + 8 * j;
        const bool in = kin && r < rows;
        ra[j] = (in && a) ? ld4(a + (size_t)(m0 + r) * a_pitch + k, a_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (amask) rm[j] = in ? *(const float4*)(amask + (size_t)(m0 + r) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (PMAX && v) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int p = 0; p < (PMAX ? PMAX : 1); ++p)
            rv[e][p] = (kin && p < P) ? v[(size_t)(k + e) * d.v_sk + (size_t)p * d.v_sp] : 0.f;
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
    auto store_chunk = [&](int k0) {
      const int k = k0 + 4 * c4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + 8 * j;
        float4 t = ra[j];
        if (PMAX && v) {
          float u[PMAX ? PMAX : 1];
#pragma unroll
          for (int p = 0; p < (PMAX ? PMAX : 1); p += 4) {
            const float4 uq = *(const float4*)(U + r * RG_UP + p);
            u[p] = uq.x; u[p + 1] = uq.y; u[p + 2] = uq.z; u[p + 3] = uq.w;
          }
#pragma unroll
          for (int p = 0; p < (PMAX ? PMAX : 1); ++p) {
            t.x = fmaf(u[p], rv[0][p], t.x);
            t.y = fmaf(u[p], rv[1][p], t.y);
            t.z = fmaf(u[p], rv[2][p], t.z);
            t.w = fmaf(u[p], rv[3][p], t.w);
          }
        }
        if (amask) { t.x *= rm[j].x; t.y *= rm[j].y; t.z *= rm[j].z; t.w *= rm[j].w; }
        *(float4*)(As + r * RG_PA + 4 * c4) = t;
        if (a_out && r < rows && k < K) *(float4*)(a_out + (size_t)(m0 + r) * K + k) = t;
      }
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
      store_chunk(k0);
      __syncthreads();
      if (k0 + RG_KC < K) load_chunk(k0 + RG_KC);
      if (wave_on) {
        const float* ap = As + (wave * 16 + l15) * RG_PA + kq;
        const float* bp = Ws + l15 * RG_PA + kq;
#pragma unroll 8
        for (int ks = 0; ks < RG_KC / 4; ks += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks], bp[4 * ks], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks + 4], bp[4 * ks + 4], acc1, 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: lane = column n0 + (lane & 15), rows 16 wave + 4 (lane >> 4) + i ------------------------------------------
  if (!wave_on) return;
  const int n = n0 + l15;
  if (n >= d.N) return;
  const float bv = d.bias ? d.bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 16 + 4 * kq + i;
    if (r >= rows) continue;
    const size_t m = (size_t)(m0 + r);
    float val = acc0[i] + acc1[i] + bv;
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
  int gx = 1, gz = 1, pmax = 0;
  size_t lds_floats = RG_TILE_FLOATS;
  for (int i = 0; i < n; ++i) {
    const hrp_regressor_step_desc& d = descs[i];
    HRP_REQUIRE(d.M > 0 && d.N >= 0 && d.P >= 0 && d.P <= HRP_REG_MAX_P, "regressor_step[%d]: M %d, N %d, P %d", i, d.M, d.N, d.P);
    if (d.P > 0) {
      HRP_REQUIRE(d.u_prev, "regressor_step[%d]: a state of width %d needs u_prev", i, d.P);
      if (d.z) {
        HRP_REQUIRE(d.zw && d.z_len > 0 && d.z_len % 16 == 0 && d.z_pitch % 4 == 0 && (uintptr_t)d.z % 16 == 0,
                    "regressor_step[%d]: z needs zw, a length that is a multiple of 16 and 16-byte aligned rows", i);
        if ((size_t)d.P * d.z_len > lds_floats) lds_floats = (size_t)d.P * d.z_len;
      }
    }
    if (d.N > 0) {
      HRP_REQUIRE(d.w && d.out && d.K > 0 && d.K % 4 == 0 && d.out_pitch >= d.N, "regressor_step[%d]: product needs w, out, K %% 4 == 0", i);
      HRP_REQUIRE(d.a || (d.v && d.P > 0), "regressor_step[%d]: no row operand", i);
      HRP_REQUIRE(d.w_sn == 1 || d.w_sk == 1, "regressor_step[%d]: the weight must be contiguous along n or along k", i);
      HRP_REQUIRE(!d.a_mask || (uintptr_t)d.a_mask % 16 == 0, "regressor_step[%d]: a_mask alignment", i);
      HRP_REQUIRE(!d.a_out || (uintptr_t)d.a_out % 16 == 0, "regressor_step[%d]: a_out alignment", i);
      HRP_REQUIRE(!d.a2 || d.w2, "regressor_step[%d]: a2 without w2", i);
      if ((d.N + RG_COLS - 1) / RG_COLS > gx) gx = (d.N + RG_COLS - 1) / RG_COLS;
    }
    if ((d.M + RG_ROWS - 1) / RG_ROWS > gz) gz = (d.M + RG_ROWS - 1) / RG_ROWS;
    if (d.P > pmax) pmax = d.P;
    args.d[i] = d;
  }
  const size_t lds = (RG_ROWS * RG_UP + lds_floats) * sizeof(float);
  HRP_REQUIRE(lds <= 160 * 1024, "regressor_step: state width x z_len does not fit LDS (%zu bytes)", lds);
  const dim3 grid(gx, n, gz), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define HRP_REG_CASE(PM)                                                                                                          \
  {                                                                                                                               \
    static bool raised = false;                                                                                                   \
    if (!raised) {                                                                                                                \
      (void)hipFuncSetAttribute((const void*)regressor_step_kernel<PM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
      raised = true;                                                                                                              \
    }                                                                                                                             \
    hipLaunchKernelGGL((regressor_step_kernel<PM>), grid, blk, lds, s, args);                                                     \
  }
  if (pmax == 0) HRP_REG_CASE(0)
  else if (pmax <= 8) HRP_REG_CASE(8)
  else HRP_REG_CASE(16)
#undef HRP_REG_CASE
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
