// Library plumbing + the small HBM-bound kernels: layout change at the stem, weight packing,
// column sums (bias gradients), batch-norm bookkeeping tables, global average pooling.
#include "hrp_common.h"
#include <type_traits>
#include <string.h>

namespace hrp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return HRP_ERR_LAUNCH;
  }
  return HRP_OK;
}

// ---- NCHW fp32 <-> NHWC T ----------------------------------------------------------------------
// One workgroup handles 64 pixels x all channels through an LDS transpose so both sides are coalesced.
// S = float: the reference's fp32 image in [0,1]; S = uint8_t: the dataset's bytes, divided by `div` on the way
// (lib/core/function.py:26,29 `.float() / 255.` on the device: ATen's GPU division by a host scalar multiplies by
// the fp32 reciprocal, and so does this kernel, so the values are bit-identical to the reference's GPU path).
template <typename T, typename S>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const S* __restrict__ src, void* __restrict__ dst,
                                                           int C, int HW, int pitch, float div) {
  // small-C path (stem: C = 3): thread per pixel, writes `pitch` channels (zero padded)
  size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  int n = blockIdx.y;
  if (p >= (size_t)HW) return;
  const S* s = src + (size_t)n * C * HW + p;
  size_t o = ((size_t)n * HW + p) * pitch;
  if (pitch * Elem<T>::SZ == 16) {      // the stem's padded pixel is one 16-byte vector: one coalesced store per thread
    constexpr int V = 16 / Elem<T>::SZ;
    float v[V];
#pragma unroll
    for (int c = 0; c < V; ++c) {
      v[c] = c < C ? (float)s[(size_t)c * HW] : 0.f;
      if (sizeof(S) == 1) v[c] = v[c] * (1.0f / div);
    }
    *(uint4*)((char*)dst + o * Elem<T>::SZ) = Elem<T>::pack(v);
    return;
  }
  for (int c = 0; c < pitch; ++c) {
    float v = c < C ? (float)s[(size_t)c * HW] : 0.f;
    if (sizeof(S) == 1) v = v * (1.0f / div);
    Elem<T>::st(dst, o + c, v);
  }
}

// NCHW fp32 [N,C,H,W] -> NHWC [N,H/2,W/2,pitch] with 2x2 space-to-depth: channel (dy*2+dx)*C + c of output pixel
// (y,x) = src[n, c, 2y+dy, 2x+dx] (odd H / W: the missing row / column reads as zero).  Turns the 7x7 stride-2 stem
// of ResNet into a 4x4 stride-1 convolution over 12 channels (one 16-channel chunk of the MFMA conv kernel).
template <typename T, typename S>
__global__ __launch_bounds__(256) void nchw_to_nhwc_s2d_kernel(const S* __restrict__ src, void* __restrict__ dst,
                                                               int C, int H, int W, int pitch, float div) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= (size_t)Ho * Wo) return;
  const int y = p / Wo, x = p - (size_t)y * Wo;
  const S* sn = src + (size_t)n * C * H * W;
  size_t o = ((size_t)n * Ho * Wo + p) * pitch;
  for (int k = 0; k < pitch; ++k) {
    float v = 0.f;
    if (k < 4 * C) {
      const int q = k / C, c = k - q * C, iy = 2 * y + (q >> 1), ix = 2 * x + (q & 1);
      if (iy < H && ix < W) v = (float)sn[((size_t)c * H + iy) * W + ix];
      if (sizeof(S) == 1) v = v * (1.0f / div);
    }
    Elem<T>::st(dst, o + k, v);
  }
}

// dst[i] (+)= idx[i] >= 0 ? src[idx[i]] : 0  (weight re-layouts whose source is a PyTorch-shaped parameter)
__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                         float* __restrict__ dst, int n, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int j = idx[i];
  const float v = j >= 0 ? src[j] : 0.f;
  dst[i] = accumulate ? dst[i] + v : v;
}

// ---- max pool 3x3, stride 2, padding 1 (ResNet stem, Resnet.py:25) ----------------------------------------
// forward: thread = (output pixel, 16-byte channel vector); `arg` (optional) keeps the window position 0..8 of the
// first maximum in scan order, which is the element PyTorch routes the gradient to.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const void* __restrict__ x, int H, int W, int C, int pitch,
                                                          void* __restrict__ y, int Ho, int Wo, int y_pitch,
                                                          uint8_t* __restrict__ arg, long total) {
  constexpr int V = Elem<T>::VEC;
  const int nv = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = i % nv;
    long r = i / nv;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho;
    const long n = r / Ho;
    float best[V];
    int bi[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { best[k] = -INFINITY; bi[k] = 0; }
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        float f[V];
        Elem<T>::unpack(*(const uint4*)((const char*)x + ((((size_t)n * H + iy) * W + ix) * pitch + cv * V) * Elem<T>::SZ), f);
#pragma unroll
        for (int k = 0; k < V; ++k)
          if (f[k] > best[k]) { best[k] = f[k]; bi[k] = ky * 3 + kx; }
      }
    }
    const size_t o = (((size_t)n * Ho + oy) * Wo + ox);
    *(uint4*)((char*)y + (o * y_pitch + cv * V) * Elem<T>::SZ) = Elem<T>::pack(best);
    if (arg) {
#pragma unroll
      for (int k = 0; k < V; ++k) arg[o * C + cv * V + k] = (uint8_t)bi[k];
    }
  }
}

// backward (gather form, no atomics): input pixel (iy,ix) collects dy of the <= 4 windows that selected it
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const void* __restrict__ dy, int Ho, int Wo, int dy_pitch,
                                                          const uint8_t* __restrict__ arg, void* __restrict__ dx, int H, int W,
                                                          int C, int pitch, int accumulate, long total) {
  constexpr int V = Elem<T>::VEC;
  const int nv = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = i % nv;
    long r = i / nv;
    const int ix = r % W; r /= W;
    const int iy = r % H;
    const long n = r / H;
    float g[V];
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = 0.f;
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {   // windows [2 oy - 1, 2 oy + 1] that contain iy
      if (oy < 0 || oy >= Ho) continue;
      const int ky = iy - (2 * oy - 1);
      if (ky < 0 || ky > 2) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        const int kx = ix - (2 * ox - 1);
        if (kx < 0 || kx > 2) continue;
        const size_t o = (((size_t)n * Ho + oy) * Wo + ox);
        float f[V];
        Elem<T>::unpack(*(const uint4*)((const char*)dy + (o * dy_pitch + cv * V) * Elem<T>::SZ), f);
        const uint8_t* a = arg + o * C + cv * V;
#pragma unroll
        for (int k = 0; k < V; ++k)
          if (a[k] == ky * 3 + kx) g[k] += f[k];
      }
    }
    char* dst = (char*)dx + ((((size_t)n * H + iy) * W + ix) * pitch + cv * V) * Elem<T>::SZ;
    if (accumulate) {
      float old[V];
      Elem<T>::unpack(*(const uint4*)dst, old);
#pragma unroll
      for (int k = 0; k < V; ++k) g[k] += old[k];
    }
    *(uint4*)dst = Elem<T>::pack(g);
  }
}

template <typename T, bool GRAD>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int pitch) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {  // r = pixel, tx = channel
    int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (p < HW && c < C) ? Elem<T>::ld(src, ((size_t)n * HW + p) * pitch + c) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {  // r = channel, tx = pixel
    int c = c0 + r, p = p0 + tx;
    if (p < HW && c < C) dst[((size_t)n * C + c) * HW + p] = tile[tx][r];
  }
}

// ---- weight packing ------------------------------------------------------------------------------
// One thread per 32-byte row of a packing (CK consecutive input channels - resp. output channels for the transposed packing -
// of one (chunk, tap, channel)); the tap is the fastest thread index so that a wave reads whole contiguous spans of the
// PyTorch-shaped source ([Cout][Cin][taps]) between its CK loads.  32-bit index arithmetic, one vector store per row (the
// element-per-thread version with 64-bit divisions took 0.46 ms for the 29 M parameters of an HRNet-W32: 2.5 % of a step).
// (blk, nblk): this workgroup's index among the nblk workgroups that share entry e
template <typename T>
__device__ __forceinline__ void pack_entry(const hrp_pack_entry& e, const unsigned blk, const unsigned nblk) {
  constexpr int CK = 32 / Elem<T>::SZ;  // one 32-byte K chunk of the conv kernels (csrc/conv_fwd.hip ROW)
  const unsigned Cout = e.Cout, Cin = e.Cin, nt = e.ntaps;
  const unsigned cout_pad = (Cout + 31) / 32 * 32, cin_pad = (Cin + 31) / 32 * 32;
  auto store_row = [&](void* dst, size_t row, const float (&v)[CK]) {
    if constexpr (std::is_same<T, f32x3_t>::value) {      // HRP_F32X3: the row's 8 values as [8 hi | 8 lo] bf16
      uint4* q = (uint4*)((char*)dst + row * 32);
      split_bf16x8(v, q[0], q[1]);
      return;
    }
    float a[CK / 2], b[CK / 2];
#pragma unroll
    for (int k = 0; k < CK / 2; ++k) { a[k] = v[k]; b[k] = v[CK / 2 + k]; }
    uint4* q = (uint4*)((char*)dst + row * 32);
    q[0] = Elem<T>::pack(a);
    q[1] = Elem<T>::pack(b);
  };
  if (e.dst) {
    const unsigned nch = (Cin + CK - 1) / CK, rows = nch * nt * cout_pad;
    for (unsigned i = blk * 256 + threadIdx.x; i < rows; i += nblk * 256) {
      const unsigned tap = i % nt, q = i / nt, co = q % cout_pad, ch = q / cout_pad;
      float v[CK];
#pragma unroll
      for (int k = 0; k < CK; ++k) {
        const unsigned ci = ch * CK + k;
        v[k] = (co < Cout && ci < Cin) ? e.src[((size_t)co * Cin + ci) * nt + tap] : 0.f;
      }
      store_row(e.dst, (size_t)(ch * nt + tap) * cout_pad + co, v);
    }
  }
  if (e.dst_t) {
    const unsigned nch = (Cout + CK - 1) / CK, rows = nch * nt * cin_pad;
    for (unsigned i = blk * 256 + threadIdx.x; i < rows; i += nblk * 256) {
      const unsigned tap = i % nt, q = i / nt, ci = q % cin_pad, ch = q / cin_pad;
      float v[CK];
#pragma unroll
      for (int k = 0; k < CK; ++k) {
        const unsigned co = ch * CK + k;
        v[k] = (co < Cout && ci < Cin) ? e.src[((size_t)co * Cin + ci) * nt + tap] : 0.f;
      }
      store_row(e.dst_t, (size_t)(ch * (nt + e.pad_t) + tap) * cin_pad + ci, v);
    }
  }
}

// one entry per blockIdx.y, the same number of workgroups for every entry (single tensors: tests, tools)
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const hrp_pack_entry* __restrict__ table) {
  const hrp_pack_entry e = table[blockIdx.y];
  pack_entry<T>(e, blockIdx.x, gridDim.x);
}

// One read of the source for both packings: a workgroup takes a tile of 32 output x 32 input channels x all taps through LDS
// (32 contiguous spans of 32 * taps floats), then writes the tile's rows of the forward packing (thread = (chunk, tap, co), co
// fastest: consecutive 32-byte rows) and of the transposed one (thread = (chunk, tap, ci)).  The row-per-thread body above reads the
// source once per packing, the transposed gather with 2.7 x over-fetch: 607 MB of fetches for the 228 MB of an HRNet pair's
// parameters.  Layers of up to PACK_TILE_TAPS taps (1 x 1, 3 x 3); the 16-tap stem forms keep the body above.
constexpr int PACK_TILE_TAPS = 9;
template <typename T>
__device__ __forceinline__ void pack_store_row(void* dst, const size_t row, const float (&v)[32 / Elem<T>::SZ]) {
  constexpr int CK = 32 / Elem<T>::SZ;
  uint4* q = (uint4*)((char*)dst + row * 32);
  if constexpr (std::is_same<T, f32x3_t>::value) {
    split_bf16x8(v, q[0], q[1]);
  } else {
    float a[CK / 2], b[CK / 2];
#pragma unroll
    for (int k = 0; k < CK / 2; ++k) { a[k] = v[k]; b[k] = v[CK / 2 + k]; }
    q[0] = Elem<T>::pack(a);
    q[1] = Elem<T>::pack(b);
  }
}

template <typename T>
__device__ __forceinline__ void pack_entry_tile(const hrp_pack_entry& e, const unsigned tile_id, float* __restrict__ lds) {
  constexpr int CK = 32 / Elem<T>::SZ, NC = 32 / CK;       // chunks of a 32-channel tile side
  const unsigned Cout = e.Cout, Cin = e.Cin, nt = e.ntaps;
  const unsigned cout_pad = (Cout + 31) / 32 * 32, cin_pad = (Cin + 31) / 32 * 32;
  const unsigned n_cit = cin_pad / 32;
  if (tile_id >= (cout_pad / 32) * n_cit) return;          // (a table built with more workgroups than hrp_pack_blocks asked for)
  const unsigned co0 = (tile_id / n_cit) * 32, ci0 = (tile_id % n_cit) * 32;
  const unsigned span = 32 * nt, pitch = span + 1;         // LDS row = one output channel: [ci local][tap], odd pitch
  // wave w takes the rows w, w + 4, ..: a row is ONE contiguous span of the source, read lane-linear; four rows' loads in flight
  {
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned valid = Cin > ci0 ? (Cin - ci0 < 32 ? (Cin - ci0) * nt : span) : 0;      // floats of a row inside the tensor
    for (unsigned r0 = wave; r0 < 32; r0 += 16) {
      for (unsigned j = lane; j < span; j += 64) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned co = co0 + r0 + 4 * u;
          v[u] = (co < Cout && j < valid) ? e.src[((size_t)co * Cin + ci0) * nt + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) lds[(r0 + 4 * u) * pitch + j] = v[u];
      }
    }
  }
  __syncthreads();
  if (e.dst) {
    const unsigned nch = (Cin + CK - 1) / CK;
    for (unsigned r = threadIdx.x; r < NC * nt * 32; r += 256) {
      const unsigned co = r & 31, q = r >> 5, tap = q % nt, c = q / nt, ch = ci0 / CK + c;
      if (ch >= nch) continue;
      float v[CK];
#pragma unroll
      for (int k = 0; k < CK; ++k) v[k] = lds[co * pitch + (c * CK + k) * nt + tap];
      pack_store_row<T>(e.dst, (size_t)(ch * nt + tap) * cout_pad + co0 + co, v);
    }
  }
  if (e.dst_t) {
    const unsigned nch = (Cout + CK - 1) / CK;
    for (unsigned r = threadIdx.x; r < NC * nt * 32; r += 256) {
      const unsigned ci = r & 31, q = r >> 5, tap = q % nt, c = q / nt, ch = co0 / CK + c;
      if (ch >= nch) continue;
      float v[CK];
#pragma unroll
      for (int k = 0; k < CK; ++k) v[k] = lds[(c * CK + k) * pitch + ci * nt + tap];
      pack_store_row<T>(e.dst_t, (size_t)(ch * (nt + e.pad_t) + tap) * cin_pad + ci0 + ci, v);
    }
  }
}

// A network's table: entry i owns the workgroups first[i] .. first[i + 1] - 1 (hrp_pack_blocks each).  The rectangular grid above
// launched 256 x 650 workgroups for the benchmark network, nearly all of them empty: 0.4 ms of workgroup dispatch that the trunks'
// first kernels on the other streams queued behind (the step lost 0.8 ms to a 0.9 GB side-stream copy).
constexpr int PACK_ROWS_PER_BLOCK = 256 * 4;
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_compact_kernel(const hrp_pack_entry* __restrict__ table, const int32_t* __restrict__ first,
                                                                   const int count) {
  int lo = 0, hi = count;                   // largest lo with first[lo] <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const hrp_pack_entry e = table[lo];
  if (e.ntaps <= PACK_TILE_TAPS) {            // (workgroup-uniform) hrp_pack_blocks gave this entry one workgroup per tile
    __shared__ float lds[32 * (32 * PACK_TILE_TAPS + 1)];
    pack_entry_tile<T>(e, blockIdx.x - first[lo], lds);
    return;
  }
  pack_entry<T>(e, blockIdx.x - first[lo], first[lo + 1] - first[lo]);
}

// ---- column sums -----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, long rows, int C, int pitch,
                                                     float* __restrict__ out, float* __restrict__ ws) {
  // thread (tx = channel within a 64-wide slab, ty = row phase); grid.x strides rows, grid.y = channel slab
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + tx;
  float s = 0.f;
  if (c < C)
    for (long r = (long)blockIdx.x * 4 + ty; r < rows; r += (long)gridDim.x * 4) s += Elem<T>::ld(x, (size_t)r * pitch + c);
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) {
    const float v = part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx];
    if (ws) ws[(size_t)blockIdx.x * C + c] = v;          // deterministic: colsum_fold_kernel adds the row groups in order
    else atomicAdd(&out[c], v);
  }
}

__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ ws, int groups, int C, float* __restrict__ out,
                                                          int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float v = 0.f;
  for (int g0 = 0; g0 < groups; g0 += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = g0 + u < groups ? ws[(size_t)(g0 + u) * C + c] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  out[c] = accumulate ? out[c] + v : v;
}

// ---- batch-norm tables -----------------------------------------------------------------------------
__global__ void bn_running_update_kernel(const hrp_bn_entry* __restrict__ table) {
  const hrp_bn_entry e = table[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
    float mean = slot_sum(e.stats, c, 2 * e.C) / e.count;
    float var = fmaxf(slot_sum(e.stats, e.C + c, 2 * e.C) / e.count - mean * mean, 0.f);
    float unbiased = e.count > 1.f ? var * (e.count / (e.count - 1.f)) : var;
    e.a[c] = (1.f - e.momentum) * e.a[c] + e.momentum * mean;
    e.b[c] = (1.f - e.momentum) * e.b[c] + e.momentum * unbiased;
  }
  if (threadIdx.x == 0 && e.counter) *e.counter += 1;
}

__global__ void bn_fold_kernel(const hrp_bn_entry* __restrict__ table) {
  const hrp_bn_entry e = table[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
    float sc = e.a[c] * rsqrtf(e.d[c] + e.eps);
    e.out_scale[c] = sc;
    e.out_shift[c] = e.b[c] - e.c[c] * sc;
  }
}

// backward sums [2C] = (sum g, sum g*xhat) -> dbeta, dgamma
__global__ void bn_param_grad_kernel(const hrp_bn_entry* __restrict__ table) {
  const hrp_bn_entry e = table[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
    float dbeta = slot_sum(e.stats, c, 2 * e.C), dgamma = slot_sum(e.stats, e.C + c, 2 * e.C);
    if (e.accumulate) { e.a[c] += dgamma; e.b[c] += dbeta; }
    else { e.a[c] = dgamma; e.b[c] = dbeta; }
  }
}

// ---- global average pool ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const void* __restrict__ x, int HW, int C, int pitch,
                                                          float* __restrict__ out, int out_pitch) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) s += Elem<T>::ld(x, ((size_t)n * HW + p) * pitch + c);
  out[(size_t)n * out_pitch + c] = s / (float)HW;
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dout, int dout_pitch, void* __restrict__ dx,
                                                          int HW, int C, int pitch, int accumulate) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float g = dout[(size_t)n * dout_pitch + c] / (float)HW;
  for (int p = 0; p < HW; ++p) {
    size_t o = ((size_t)n * HW + p) * pitch + c;
    Elem<T>::st(dx, o, accumulate ? Elem<T>::ld(dx, o) + g : g);
  }
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int sp, float* __restrict__ dst, int dp, int rows, int cols, int acc) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  int r = i / cols, c = i % cols;
  float v = src[(size_t)r * sp + c];
  float* d = dst + (size_t)r * dp + c;
  *d = acc ? *d + v : v;
}

// up to HRP_COPY_MAX column-block copies (or zero fills: src == NULL) in one launch: blockIdx.y = problem
struct CopyBatchArgs { hrp_copy_desc d[HRP_COPY_MAX]; };
__global__ void copy_cols_batch_kernel(const CopyBatchArgs a) {
  const hrp_copy_desc& d = a.d[blockIdx.y];
  const size_t total = (size_t)d.rows * d.cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / d.cols), c = (int)(i - (size_t)r * d.cols);
    const float v = d.src ? d.src[(size_t)r * d.src_pitch + c] : 0.f;
    float* q = d.dst + (size_t)r * d.dst_pitch + c;
    *q = d.accumulate ? *q + v : v;
  }
}

__global__ void scale_rows_kernel(float* __restrict__ x, int pitch, int rows, int cols, const float* __restrict__ rs, float s) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  int r = i / cols, c = i % cols;
  x[(size_t)r * pitch + c] *= (rs ? rs[r] : 1.f) * s;
}

__global__ void mul_kernel(const float* __restrict__ x, int xp, const float* __restrict__ m, int mp, float* __restrict__ y, int yp,
                           int rows, int cols, int acc) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  int r = i / cols, c = i % cols;
  float v = x[(size_t)r * xp + c] * m[(size_t)r * mp + c];
  float* d = y + (size_t)r * yp + c;
  *d = acc ? *d + v : v;
}

// ---- dropout of the regression heads (reference full_net.py:98-99, 132-133 nn.Dropout(p)) -------------------------
// Philox4x32-10 (Salmon et al., the generator torch / cuRAND use): key = the plan's seed, counter = (element / 4, salt of
// the dropout op, step).  Everything the kernel needs lives on the device - the seed and a step counter one launch per
// forward advances - so the mask is drawn on the stream the consumer runs on and a captured HIP graph draws a fresh
// mask every replay.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void rng_advance_kernel(uint64_t* state) { state[1] += 1; }

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int xp, float* __restrict__ y, int yp,
                                                      float* __restrict__ mask, int rows, int cols, float keep,
                                                      const uint64_t* __restrict__ state, uint32_t salt) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;     // 4 consecutive elements of the [rows, cols] mask
  const size_t n = (size_t)rows * cols;
  if (q * 4 >= n) return;
  const uint64_t seed = state[0], step = state[1];
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32) ^ salt, (uint32_t)step, (uint32_t)(step >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float scale = 1.0f / keep;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t i = q * 4 + j;
    if (i >= n) break;
    const float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);     // uniform in [0, 1)
    const float m = u < keep ? scale : 0.f;
    const int r = (int)(i / cols), cc = (int)(i - (size_t)r * cols);
    mask[i] = m;
    y[(size_t)r * yp + cc] = x[(size_t)r * xp + cc] * m;
  }
}

}  // namespace hrp

using namespace hrp;

#ifndef HRP_SRC_HASH
#define HRP_SRC_HASH "unknown"
#endif
extern "C" const char* hrp_source_hash(void) { return HRP_SRC_HASH; }

extern "C" int hrp_rng_advance(uint64_t* state_dev, void* stream) {
  HRP_REQUIRE(state_dev, "rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev);
  return check_launch("rng_advance");
}

extern "C" int hrp_dropout_f32(const float* x, int x_pitch, float* y, int y_pitch, float* mask, int rows, int cols, float keep,
                               const uint64_t* state_dev, uint32_t salt, void* stream) {
  HRP_REQUIRE(x && y && mask && state_dev && rows > 0 && cols > 0, "dropout: bad args");
  HRP_REQUIRE(keep > 0.f && keep <= 1.f, "dropout: keep probability %f", (double)keep);
  const size_t quads = ((size_t)rows * cols + 3) / 4;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x_pitch, y, y_pitch,
                     mask, rows, cols, keep, state_dev, salt);
  return check_launch("dropout");
}

extern "C" int hrp_mul_f32(const float* x, int x_pitch, const float* m, int m_pitch, float* y, int y_pitch, int rows, int cols,
                           int accumulate, void* stream) {
  HRP_REQUIRE(x && m && y && rows > 0 && cols > 0, "mul_f32: bad args");
  hipLaunchKernelGGL(mul_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, x, x_pitch, m, m_pitch, y, y_pitch,
                     rows, cols, accumulate);
  return check_launch("mul_f32");
}

extern "C" const char* hrp_last_error(void) { return hrp::g_err; }
extern "C" int hrp_version(void) { return 100; }

extern "C" int hrp_device_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

extern "C" int hrp_nchw_to_nhwc(const float* src, void* dst, int dtype, int N, int C, int H, int W, int dst_pitch, void* stream) {
  HRP_REQUIRE(src && dst && N > 0 && C > 0 && dst_pitch >= C, "nchw_to_nhwc: bad args");
  dim3 grid(cdiv(H * W, 256), N);
  if (dtype == HRP_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, H * W, dst_pitch, 1.f);
  else hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t, float>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, H * W, dst_pitch, 1.f);
  return check_launch("nchw_to_nhwc");
}

extern "C" int hrp_u8_nchw_to_nhwc(const uint8_t* src, void* dst, int dtype, int N, int C, int H, int W, int dst_pitch,
                                   float divisor, int s2d, void* stream) {
  HRP_REQUIRE(src && dst && N > 0 && C > 0 && divisor > 0.f, "u8_nchw_to_nhwc: bad args");
  HRP_REQUIRE(dst_pitch >= (s2d ? 4 * C : C), "u8_nchw_to_nhwc: dst_pitch too small");
  hipStream_t st = (hipStream_t)stream;
  if (s2d) {
    dim3 grid(cdiv(((H + 1) / 2) * ((W + 1) / 2), 256), N);
    if (dtype == HRP_F32) hipLaunchKernelGGL((nchw_to_nhwc_s2d_kernel<float, uint8_t>), grid, dim3(256), 0, st, src, dst, C, H, W, dst_pitch, divisor);
    else hipLaunchKernelGGL((nchw_to_nhwc_s2d_kernel<bf16_t, uint8_t>), grid, dim3(256), 0, st, src, dst, C, H, W, dst_pitch, divisor);
  } else {
    dim3 grid(cdiv(H * W, 256), N);
    if (dtype == HRP_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, uint8_t>), grid, dim3(256), 0, st, src, dst, C, H * W, dst_pitch, divisor);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t, uint8_t>), grid, dim3(256), 0, st, src, dst, C, H * W, dst_pitch, divisor);
  }
  return check_launch("u8_nchw_to_nhwc");
}

extern "C" int hrp_nchw_to_nhwc_s2d(const float* src, void* dst, int dtype, int N, int C, int H, int W, int dst_pitch, void* stream) {
  HRP_REQUIRE(src && dst && N > 0 && C > 0 && dst_pitch >= 4 * C, "nchw_to_nhwc_s2d: bad args");
  dim3 grid(cdiv(((H + 1) / 2) * ((W + 1) / 2), 256), N);
  if (dtype == HRP_F32) hipLaunchKernelGGL((nchw_to_nhwc_s2d_kernel<float, float>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, H, W, dst_pitch, 1.f);
  else hipLaunchKernelGGL((nchw_to_nhwc_s2d_kernel<bf16_t, float>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, H, W, dst_pitch, 1.f);
  return check_launch("nchw_to_nhwc_s2d");
}

extern "C" int hrp_fill_zero(void* p, int64_t bytes, void* stream) {
  HRP_REQUIRE(p && bytes >= 0, "fill_zero: bad args");
  zero_async(p, (size_t)bytes, (hipStream_t)stream);
  return check_launch("fill_zero");
}

extern "C" int hrp_gather_f32(const float* src, const int32_t* idx, float* dst, int n, int accumulate, void* stream) {
  HRP_REQUIRE(src && idx && dst && n > 0, "gather_f32: bad args");
  hipLaunchKernelGGL(gather_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, accumulate);
  return check_launch("gather_f32");
}

extern "C" int hrp_maxpool3x3s2_fwd(const void* x, int dtype, int N, int H, int W, int C, int pitch, void* y, int y_pitch,
                                    uint8_t* argmax, void* stream) {
  const int vec = dtype == HRP_F32 ? 4 : 8, sz = dtype == HRP_F32 ? 4 : 2;
  HRP_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "maxpool_fwd: bad args");
  HRP_REQUIRE(C % vec == 0 && pitch % vec == 0 && y_pitch % vec == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0,
              "maxpool_fwd: channels / pitches must be multiples of %d and the tensors 16-byte aligned", vec);
  (void)sz;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / vec);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dtype == HRP_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, H, W, C, pitch, y, Ho, Wo, y_pitch, argmax, total);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, H, W, C, pitch, y, Ho, Wo, y_pitch, argmax, total);
  return check_launch("maxpool_fwd");
}

extern "C" int hrp_maxpool3x3s2_bwd(const void* dy, int dy_pitch, const uint8_t* argmax, void* dx, int dtype, int N, int H, int W,
                                    int C, int pitch, int accumulate, void* stream) {
  const int vec = dtype == HRP_F32 ? 4 : 8;
  HRP_REQUIRE(dy && argmax && dx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool_bwd: bad args");
  HRP_REQUIRE(C % vec == 0 && pitch % vec == 0 && dy_pitch % vec == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)dy % 16 == 0,
              "maxpool_bwd: channels / pitches must be multiples of %d and the tensors 16-byte aligned", vec);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * H * W * (C / vec);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dtype == HRP_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dy, Ho, Wo, dy_pitch, argmax, dx, H, W, C, pitch, accumulate, total);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dy, Ho, Wo, dy_pitch, argmax, dx, H, W, C, pitch, accumulate, total);
  return check_launch("maxpool_bwd");
}

extern "C" int hrp_nhwc_to_nchw(const void* src, float* dst, int dtype, int N, int C, int H, int W, int src_pitch, void* stream) {
  HRP_REQUIRE(src && dst && N > 0 && C > 0 && src_pitch >= C, "nhwc_to_nchw: bad args");
  dim3 grid(cdiv(H * W, 64), cdiv(C, 64), N);
  if (dtype == HRP_F32) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, H * W, src_pitch);
  else hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, H * W, src_pitch);
  return check_launch("nhwc_to_nchw");
}

extern "C" int hrp_nchw_grad_from_nhwc(const void* src, float* dst, int dtype, int N, int C, int H, int W, int src_pitch, void* stream) {
  return hrp_nhwc_to_nchw(src, dst, dtype, N, C, H, W, src_pitch, stream);
}

extern "C" int hrp_pack_weights(const hrp_pack_entry* table_dev, int count, int dtype, int max_elems, void* stream) {
  HRP_REQUIRE(table_dev && count > 0, "pack_weights: empty table");
  int bx = cdiv(max_elems, 256 * 16 * 4);      // 256 threads x 16-element rows x ~4 rows per thread for the largest tensor
  if (bx < 1) bx = 1;
  if (bx > 256) bx = 256;                      // (workgroups past a small tensor's rows exit at once)
  dim3 grid(bx, count);
  if (dtype == HRP_F32) hipLaunchKernelGGL(pack_weights_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, table_dev);
  else if (dtype == HRP_F32X3) hipLaunchKernelGGL(pack_weights_kernel<f32x3_t>, grid, dim3(256), 0, (hipStream_t)stream, table_dev);
  else hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, table_dev);
  return check_launch("pack_weights");
}

extern "C" int hrp_pack_blocks(int Cout, int Cin, int ntaps, int dtype, int has_dst, int has_dst_t) {
  if (Cout <= 0 || Cin <= 0 || ntaps <= 0) return 0;
  const int CK = dtype == HRP_BF16 ? 16 : 8;
  const long cout_pad = (Cout + 31) / 32 * 32, cin_pad = (Cin + 31) / 32 * 32;
  if (ntaps <= PACK_TILE_TAPS) return (int)((cout_pad / 32) * (cin_pad / 32));      // pack_entry_tile: one workgroup per 32 x 32 tile
  const long rf = has_dst ? (long)cdiv(Cin, CK) * ntaps * cout_pad : 0, rt = has_dst_t ? (long)cdiv(Cout, CK) * ntaps * cin_pad : 0;
  const long rows = rf > rt ? rf : rt;      // (a workgroup walks its share of the forward rows, then of the transposed rows)
  const long b = (rows + PACK_ROWS_PER_BLOCK - 1) / PACK_ROWS_PER_BLOCK;
  return (int)(b < 1 ? 1 : b);
}

extern "C" int hrp_pack_weights_compact(const hrp_pack_entry* table_dev, const int32_t* first_block_dev, int count, int total_blocks,
                                        int dtype, void* stream) {
  HRP_REQUIRE(table_dev && first_block_dev && count > 0 && total_blocks >= count, "pack_weights_compact: empty table / fewer workgroups than entries");
  const dim3 grid(total_blocks);
  if (dtype == HRP_F32) hipLaunchKernelGGL(pack_weights_compact_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, table_dev, first_block_dev, count);
  else if (dtype == HRP_F32X3) hipLaunchKernelGGL(pack_weights_compact_kernel<f32x3_t>, grid, dim3(256), 0, (hipStream_t)stream, table_dev, first_block_dev, count);
  else hipLaunchKernelGGL(pack_weights_compact_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, table_dev, first_block_dev, count);
  return check_launch("pack_weights_compact");
}

static inline int colsum_groups(int64_t rows) {
  int gx = (int)((rows + 63) / 64);
  if (gx > 512) gx = 512;
  return gx < 1 ? 1 : gx;
}

extern "C" int64_t hrp_colsum_workspace_bytes(int64_t rows, int C) { return rows > 0 && C > 0 ? (int64_t)colsum_groups(rows) * C * 4 : 0; }

extern "C" int hrp_colsum(const void* x, int dtype, int64_t rows, int C, int pitch, float* out, int accumulate, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  HRP_REQUIRE(x && out && rows > 0 && C > 0, "colsum: bad args");
  const int gx = colsum_groups(rows);
  float* ws = (float*)workspace;
  HRP_REQUIRE(!ws || workspace_bytes >= (int64_t)gx * C * 4, "colsum: workspace too small");
  if (!ws && !accumulate) zero_async(out, sizeof(float) * C, (hipStream_t)stream);
  dim3 grid(gx, cdiv(C, 64));
  if (dtype == HRP_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)rows, C, pitch, out, ws);
  else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)rows, C, pitch, out, ws);
  if (ws) hipLaunchKernelGGL(colsum_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, ws, gx, C, out, accumulate);
  return check_launch("colsum");
}

extern "C" int hrp_bn_running_update(const hrp_bn_entry* table_dev, int count, void* stream) {
  HRP_REQUIRE(table_dev && count > 0, "bn_running_update: empty table");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(count), dim3(256), 0, (hipStream_t)stream, table_dev);
  return check_launch("bn_running_update");
}
extern "C" int hrp_bn_fold(const hrp_bn_entry* table_dev, int count, void* stream) {
  HRP_REQUIRE(table_dev && count > 0, "bn_fold: empty table");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(count), dim3(256), 0, (hipStream_t)stream, table_dev);
  return check_launch("bn_fold");
}
extern "C" int hrp_bn_param_grad(const hrp_bn_entry* table_dev, int count, void* stream) {
  HRP_REQUIRE(table_dev && count > 0, "bn_param_grad: empty table");
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3(count), dim3(256), 0, (hipStream_t)stream, table_dev);
  return check_launch("bn_param_grad");
}

// ---- clip_grad_norm_ + Adam over all parameters, table driven -----------------------------------------
__global__ __launch_bounds__(256) void opt_grad_sumsq_kernel(const hrp_opt_tensor* __restrict__ tensors,
                                                             const hrp_opt_chunk* __restrict__ chunks,
                                                             float* __restrict__ slots, float* __restrict__ chunk_sums) {
  const hrp_opt_chunk ck = chunks[blockIdx.x];
  const hrp_opt_tensor t = tensors[ck.tensor];
  const int64_t base = (int64_t)ck.offset * HRP_OPT_CHUNK;
  const int64_t left = t.numel - base;
  const int n = left < HRP_OPT_CHUNK ? (int)left : HRP_OPT_CHUNK;
  const float* g = t.grad + base;
  float s = 0.f;
  if ((n & 3) == 0 && ((uintptr_t)g & 15) == 0) {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      const float4 v = *(const float4*)(g + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = part[0] + part[1] + part[2] + part[3];
    if (chunk_sums) chunk_sums[blockIdx.x] = tot;      // deterministic path: folded in a fixed order by opt_fold_sumsq_kernel
    else atomicAdd(&slots[blockIdx.x & (HRP_STAT_SLOTS - 1)], tot);
  }
}

// slots[0] = sum of chunk_sums[0 .. n) in a FIXED order (thread t takes t, t + 256, ..; then a fixed tree), slots[1..] = 0:
// the same bits on every data-parallel rank, whatever order the chunk kernels finished in (fp32 atomics into the slots
// made the clip coefficient - and with it every parameter - differ by an ulp between replicas holding identical gradients)
__global__ __launch_bounds__(256) void opt_fold_sumsq_kernel(const float* __restrict__ chunk_sums, int n, float* __restrict__ slots) {
  __shared__ float part[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += chunk_sums[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < HRP_STAT_SLOTS) slots[threadIdx.x] = threadIdx.x == 0 ? part[0] : 0.f;
}

__global__ __launch_bounds__(256) void opt_adam_kernel(const hrp_opt_tensor* __restrict__ tensors,
                                                       const hrp_opt_chunk* __restrict__ chunks,
                                                       const float* __restrict__ slots, float max_norm,
                                                       const float* __restrict__ step_dev, float lr, float b1, float b2,
                                                       float eps) {
  const hrp_opt_chunk ck = chunks[blockIdx.x];
  const hrp_opt_tensor t = tensors[ck.tensor];
  const int64_t base = (int64_t)ck.offset * HRP_OPT_CHUNK;
  const int64_t left = t.numel - base;
  const int n = left < HRP_OPT_CHUNK ? (int)left : HRP_OPT_CHUNK;
  float clip = 1.f;
  if (max_norm > 0.f) {
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < HRP_STAT_SLOTS; ++k) ss += slots[k];
    clip = fminf(1.f, max_norm / (sqrtf(ss) + 1e-6f));
  }
  const float step = *step_dev;
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  float* p = t.param + base;
  float* g = t.grad + base;
  float* m = t.exp_avg + base;
  float* v = t.exp_avg_sq + base;
  auto upd = [&](float& pp, float& gg, float& mm, float& vv) {
    gg *= clip;
    mm = b1 * mm + (1.f - b1) * gg;
    vv = b2 * vv + (1.f - b2) * gg * gg;
    pp -= step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
  };
  if ((n & 3) == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      float4 pp = *(float4*)(p + i), gg = *(float4*)(g + i), mm = *(float4*)(m + i), vv = *(float4*)(v + i);
      upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
      *(float4*)(p + i) = pp; *(float4*)(g + i) = gg; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
  }
}

extern "C" int hrp_opt_grad_sumsq(const hrp_opt_tensor* tensors_dev, const hrp_opt_chunk* chunks_dev, int nchunks,
                                  float* sumsq_slots, float* chunk_sums, void* stream) {
  HRP_REQUIRE(tensors_dev && chunks_dev && sumsq_slots && nchunks > 0, "opt_grad_sumsq: bad args");
  hipLaunchKernelGGL(opt_grad_sumsq_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, tensors_dev, chunks_dev, sumsq_slots,
                     chunk_sums);
  if (chunk_sums)
    hipLaunchKernelGGL(opt_fold_sumsq_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, chunk_sums, nchunks, sumsq_slots);
  return check_launch("opt_grad_sumsq");
}
extern "C" int hrp_opt_adam_step(const hrp_opt_tensor* tensors_dev, const hrp_opt_chunk* chunks_dev, int nchunks,
                                 const float* sumsq_slots, float max_norm, const float* step_dev,
                                 float lr, float beta1, float beta2, float eps, void* stream) {
  HRP_REQUIRE(tensors_dev && chunks_dev && step_dev && nchunks > 0, "opt_adam_step: bad args");
  HRP_REQUIRE(max_norm <= 0.f || sumsq_slots, "opt_adam_step: clipping needs the sum-of-squares slots");
  hipLaunchKernelGGL(opt_adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, tensors_dev, chunks_dev, sumsq_slots,
                     max_norm, step_dev, lr, beta1, beta2, eps);
  return check_launch("opt_adam_step");
}

extern "C" int hrp_avgpool_fwd(const void* x, int dtype, int N, int HW, int C, int pitch, float* out, int out_pitch, void* stream) {
  HRP_REQUIRE(x && out && N > 0 && HW > 0 && C > 0, "avgpool_fwd: bad args");
  dim3 grid(cdiv(C, 256), N);
  if (dtype == HRP_F32) hipLaunchKernelGGL(avgpool_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, HW, C, pitch, out, out_pitch);
  else hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, HW, C, pitch, out, out_pitch);
  return check_launch("avgpool_fwd");
}
extern "C" int hrp_avgpool_bwd(const float* dout, int dout_pitch, void* dx, int dtype, int N, int HW, int C, int pitch, int accumulate, void* stream) {
  HRP_REQUIRE(dout && dx && N > 0 && HW > 0 && C > 0, "avgpool_bwd: bad args");
  dim3 grid(cdiv(C, 256), N);
  if (dtype == HRP_F32) hipLaunchKernelGGL(avgpool_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, dout, dout_pitch, dx, HW, C, pitch, accumulate);
  else hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, dout, dout_pitch, dx, HW, C, pitch, accumulate);
  return check_launch("avgpool_bwd");
}

extern "C" int hrp_copy_cols(const float* src, int src_pitch, float* dst, int dst_pitch, int rows, int cols, int accumulate, void* stream) {
  HRP_REQUIRE(src && dst && rows > 0 && cols > 0, "copy_cols: bad args");
  hipLaunchKernelGGL(copy_cols_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, rows, cols, accumulate);
  return check_launch("copy_cols");
}
extern "C" int hrp_copy_cols_batch(const hrp_copy_desc* descs, int n, void* stream) {
  HRP_REQUIRE(descs && n > 0 && n <= HRP_COPY_MAX, "copy_cols_batch: 1 .. %d problems (n=%d)", HRP_COPY_MAX, n);
  CopyBatchArgs a;
  size_t most = 0;
  for (int i = 0; i < n; ++i) {
    const hrp_copy_desc& d = descs[i];
    HRP_REQUIRE(d.dst && d.rows > 0 && d.cols > 0 && d.dst_pitch >= d.cols && (!d.src || d.src_pitch >= d.cols),
                "copy_cols_batch: problem %d: bad pointers / shape (rows=%d cols=%d pitches %d -> %d)", i, d.rows, d.cols, d.src_pitch, d.dst_pitch);
    HRP_REQUIRE(d.src || !d.accumulate, "copy_cols_batch: problem %d accumulates nothing (src == NULL)", i);
    a.d[i] = d;
    const size_t t = (size_t)d.rows * d.cols;
    most = t > most ? t : most;
  }
  const int gx = (int)(most >= 64 * 256 ? 64 : cdiv((int)most, 256));
  hipLaunchKernelGGL(copy_cols_batch_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("copy_cols_batch");
}
extern "C" int hrp_scale_rows(float* x, int pitch, int rows, int cols, const float* row_scale, float s, void* stream) {
  HRP_REQUIRE(x && rows > 0 && cols > 0, "scale_rows: bad args");
  hipLaunchKernelGGL(scale_rows_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, x, pitch, rows, cols, row_scale, s);
  return check_launch("scale_rows");
}
