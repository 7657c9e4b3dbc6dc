// Shared device/host helpers for libhrp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/hrp.h"

namespace hrp {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define HRP_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      hrp::set_error(__VA_ARGS__);    \
      return HRP_ERR_ARG;             \
    }                                 \
  } while (0)

// ---- element types ---------------------------------------------------------------------------
struct bf16_t {
  uint16_t v;
};

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16 through the compiler's cast: gfx950 has v_cvt_pk_bf16_f32 (round to nearest even, two values
// per instruction); the software sequence it replaces cost ~7 VALU ops per element in every epilogue
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  hw_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int SZ = 4;
  static constexpr int VEC = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
  __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
  __device__ static __forceinline__ void unpack(const uint4& r, float* f) {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y);
    f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <>
struct Elem<bf16_t> {
  static constexpr int SZ = 2;
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const void* p, size_t i) { return bf2f(((const uint16_t*)p)[i]); }
  __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((uint16_t*)p)[i] = f2bf(v); }
  __device__ static __forceinline__ void unpack(const uint4& r, float* f) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    uint4 r;
    r.x = pack_bf2(f[0], f[1]);
    r.y = pack_bf2(f[2], f[3]);
    r.z = pack_bf2(f[4], f[5]);
    r.w = pack_bf2(f[6], f[7]);
    return r;
  }
};

// fp32 tensors, matrix products as THREE bf16 MFMAs on split operands (x = hi + lo, both bf16: x w ~ hi_x hi_w + hi_x lo_w +
// lo_x hi_w, relative error ~2^-16 per product instead of bf16's 2^-9, at a third of the bf16 matrix rate instead of fp32's
// sixteenth): element type tag of the HRP_F32X3 convolution kernels.  Everything but the MFMA operands is float.
struct f32x3_t {
  float v;
};
template <>
struct Elem<f32x3_t> : Elem<float> {};
// x[0..7] -> hi, lo (8 bf16 each, as 4 dwords): hi = rne(x), lo = rne(x - hi)
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], uint4& hi, uint4& lo) {
  hi = Elem<bf16_t>::pack(x);
  float h[8], r[8];
  Elem<bf16_t>::unpack(hi, h);
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = x[i] - h[i];
  lo = Elem<bf16_t>::pack(r);
}

// sum of the HRP_STAT_SLOTS replicas of statistic element i (buffer laid out [slot][n])
__device__ __forceinline__ float slot_sum(const double* p, int i, int n) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < HRP_STAT_SLOTS; ++k) s += p[k * n + i];
  return s;
}

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// division by a launch-invariant divisor: q = umulhi(v, m) is exact while v * d < 2^32 (m = 2^32/d + 1)
struct FastDiv {
  uint32_t m, d;
  float inv;
};
static inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = (uint32_t)d;
  f.m = d <= 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1ull);
  f.inv = 1.0f / (float)(d < 1 ? 1 : d);
  return f;
}
__device__ __forceinline__ int fdiv(int v, const FastDiv& f) { return f.m ? (int)__umulhi((uint32_t)v, f.m) : v; }
// v / d for 0 <= v < 65536 with three full-rate instructions (cvt, mul, cvt) instead of a quarter-rate
// 32-bit multiply: (v + 0.5) / d is at least 0.5 / d away from an integer and the fp32 error is below
// 2^-7 / d, so the truncation is exact (tests/test_host_cpu.py checks the same arithmetic in numpy).
__device__ __forceinline__ int fdiv16(int v, const FastDiv& f) { return (int)(((float)v + 0.5f) * f.inv); }
// a * b for 0 <= a, b < 2^24 (full-rate v_mul_u32_u24; v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ int mul24(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }

// Stream-ordered zero fill as an ordinary kernel of THIS library (16-byte aligned pointer and size: every caller
// passes fp32 / bf16 tensors whose byte size is a multiple of 16, the tail loop covers the rest).  Used instead of
// hipMemsetAsync: a memset node sits outside the kernel queue's in-order dispatch on some paths (round 2 measured lost
// split-K contributions when a memset on a side stream was followed by the kernel that accumulates into the buffer).
static __global__ void zero_fill_kernel(uint4* __restrict__ p, size_t n16, unsigned char* __restrict__ tail, int ntail) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
static inline void zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (!ptr || bytes == 0) return;
  unsigned char* b = (unsigned char*)ptr;
  size_t head = ((uintptr_t)b % 16) ? 16 - ((uintptr_t)b % 16) : 0;
  if (head > bytes) head = bytes;
  if (head) hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(64), 0, s, (uint4*)nullptr, (size_t)0, b, (int)head);
  b += head; bytes -= head;
  const size_t n16 = bytes / 16;
  const int ntail = (int)(bytes % 16);
  if (n16 == 0 && ntail == 0) return;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint4*)b, n16, b + n16 * 16, ntail);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

}  // namespace hrp
