// Device-side pieces of the segmentation-mask network of BASELINE config 5 (reference lib/models/ctrnet/mask_inference.py:44-57,
// keypoint_seg_resnet.py:134-149, CtRNet.py:102-111) that are not convolutions:
//   hrp_pil_resize_normalize   np.uint8(img) -> PIL Image.resize (bicubic, 8-bit, two passes) -> ToTensor -> Normalize, written in the
//                              ResNet stem's 2x2 space-to-depth layout (the reference: a per-image loop on the HOST)
//   hrp_broadcast_hw           ASPP image pooling: bilinear up-sampling of a 1 x 1 map = the per-image vector at every pixel
//   hrp_bilinear_nhwc_to_nchw  F.interpolate(mode='bilinear', align_corners=False) of the logits to the input size (+ sigmoid)
// All HBM / latency bound and tiny next to the trunk (a 240 x 320 mask per image).
#include "hrp_common.h"

namespace hrp {

// Pillow's resampling tables (src/libImaging/Resample.c: precompute_coeffs + normalize_coeffs_8bpc, bicubic a = -0.5): per
// output index [first input index, tap count, taps in 2^-22 units ...]; rows of HRP_PIL_KMAX + 2 ints.
static double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

// one output pixel (all three channels): the horizontal pass of the <= KMAX input rows it needs (rounded to 8 bits, as Pillow's
// intermediate image), then the vertical pass.  SRC: float (truncated to uint8 like np.uint8) or uint8, NCHW.
template <typename SRC, typename T>
__global__ void pil_resize_kernel(const SRC* __restrict__ src, const int32_t* __restrict__ xtab, const int32_t* __restrict__ ytab,
                                  T* __restrict__ dst, int H, int W, int Ho, int Wo, int dst_pitch, int s2d,
                                  float m0, float m1, float m2, float s0, float s1, float s2) {
  const int n = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= Ho * Wo) return;
  const int oy = o / Wo, ox = o - oy * Wo;
  const int32_t* xt = xtab + ox * (HRP_PIL_KMAX + 2);
  const int32_t* yt = ytab + oy * (HRP_PIL_KMAX + 2);
  const int x0 = xt[0], nx = xt[1], y0 = yt[0], ny = yt[1];
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  float outv[3];
  for (int c = 0; c < 3; ++c) {
    const SRC* pl = src + ((size_t)n * 3 + c) * H * W;
    int acc_v = 1 << (PIL_PRECISION_BITS - 1);
    for (int j = 0; j < ny; ++j) {
      const SRC* row = pl + (size_t)(y0 + j) * W + x0;
      int acc_h = 1 << (PIL_PRECISION_BITS - 1);
      for (int i = 0; i < nx; ++i) acc_h += (int)(uint8_t)(int)row[i] * xt[2 + i];
      int t = acc_h >> PIL_PRECISION_BITS;
      t = t < 0 ? 0 : (t > 255 ? 255 : t);
      acc_v += t * yt[2 + j];
    }
    int v = acc_v >> PIL_PRECISION_BITS;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    outv[c] = ((float)v / 255.0f - mean[c]) / sd[c];       // ToTensor: byte / 255; Normalize: (x - mean) / std
  }
  if (s2d) {   // dst[n, oy / 2, ox / 2, ((oy & 1) * 2 + (ox & 1)) * 3 + c]
    T* q = dst + ((size_t)n * (Ho / 2) * (Wo / 2) + (size_t)(oy >> 1) * (Wo / 2) + (ox >> 1)) * dst_pitch + ((oy & 1) * 2 + (ox & 1)) * 3;
    for (int c = 0; c < 3; ++c) Elem<T>::st(q, c, outv[c]);
  } else {
    T* q = dst + ((size_t)n * Ho * Wo + o) * dst_pitch;
    for (int c = 0; c < 3; ++c) Elem<T>::st(q, c, outv[c]);
  }
}

template <typename T>
__global__ void broadcast_hw_kernel(const float* __restrict__ src, int src_pitch, T* __restrict__ dst, int HW, int C, int dst_pitch) {
  const int n = blockIdx.y;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)HW * C) return;
  const int p = (int)(e / C), c = (int)(e - (size_t)p * C);
  Elem<T>::st(dst + ((size_t)n * HW + p) * dst_pitch, c, src[(size_t)n * src_pitch + c]);
}

// ATen upsample_bilinear2d, align_corners = False: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out
template <typename T>
__global__ void bilinear_kernel(const T* __restrict__ src, int h, int w, int C, int pitch, float* __restrict__ dst, int H, int W, int act) {
  const int n = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= H * W) return;
  const int oy = o / W, ox = o - oy * W;
  const float sy = fmaxf(((float)h / (float)H) * ((float)oy + 0.5f) - 0.5f, 0.f);
  const float sx = fmaxf(((float)w / (float)W) * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  const T* b = src + (size_t)n * h * w * pitch;
  for (int c = 0; c < C; ++c) {
    const float v00 = Elem<T>::ld(b, (size_t)(y0 * w + x0) * pitch + c), v01 = Elem<T>::ld(b, (size_t)(y0 * w + x1) * pitch + c);
    const float v10 = Elem<T>::ld(b, (size_t)(y1 * w + x0) * pitch + c), v11 = Elem<T>::ld(b, (size_t)(y1 * w + x1) * pitch + c);
    float v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    if (act == 1) v = 1.f / (1.f + expf(-v));
    dst[((size_t)n * C + c) * H * W + o] = v;
  }
}

}  // namespace hrp

using namespace hrp;

extern "C" int hrp_pil_resize_table(int in_size, int out_size, int32_t* out) {
  HRP_REQUIRE(out && in_size > 0 && out_size > 0, "pil_resize_table: bad args");
  const double scale = (double)in_size / (double)out_size;
  const double fscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * fscale;
  HRP_REQUIRE((int)(2 * support + 1.5) <= HRP_PIL_KMAX, "pil_resize_table: reduction factor %.2f needs more than %d taps", scale, HRP_PIL_KMAX);
  for (int xx = 0; xx < out_size; ++xx) {
    int32_t* row = out + (size_t)xx * (HRP_PIL_KMAX + 2);
    const double center = (xx + 0.5) * scale, ss = 1.0 / fscale;
    int lo = (int)(center - support + 0.5);
    if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5);
    if (hi > in_size) hi = in_size;
    const int n = hi - lo;
    HRP_REQUIRE(n <= HRP_PIL_KMAX, "pil_resize_table: %d taps", n);
    double w[HRP_PIL_KMAX], tot = 0.0;
    for (int x = 0; x < n; ++x) { w[x] = bicubic_filter((x + lo - center + 0.5) * ss); tot += w[x]; }
    row[0] = lo; row[1] = n;
    for (int x = 0; x < HRP_PIL_KMAX; ++x) {
      double v = x < n ? (tot != 0.0 ? w[x] / tot : w[x]) : 0.0;
      row[2 + x] = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << PIL_PRECISION_BITS)) : (int32_t)(0.5 + v * (double)(1 << PIL_PRECISION_BITS));
    }
  }
  return HRP_OK;
}

extern "C" int hrp_pil_resize_normalize(const void* src, int src_u8, int N, int H, int W, const int32_t* xtab_dev, const int32_t* ytab_dev,
                                        int Ho, int Wo, void* dst, int dtype, int dst_pitch, int s2d, const float* mean3,
                                        const float* std3, void* stream) {
  HRP_REQUIRE(src && dst && xtab_dev && ytab_dev && mean3 && std3 && N > 0 && Ho > 0 && Wo > 0, "pil_resize_normalize: bad args");
  HRP_REQUIRE(!s2d || (Ho % 2 == 0 && Wo % 2 == 0 && dst_pitch >= 12), "pil_resize_normalize: the space-to-depth layout needs even sizes and pitch >= 12");
  HRP_REQUIRE(s2d || dst_pitch >= 3, "pil_resize_normalize: dst_pitch");
  dim3 grid(cdiv(Ho * Wo, 256), N);
  hipStream_t st = (hipStream_t)stream;
#define HRP_RZ(SRC, T) hipLaunchKernelGGL((pil_resize_kernel<SRC, T>), grid, dim3(256), 0, st, (const SRC*)src, xtab_dev, ytab_dev, (T*)dst, H, W, \
                                          Ho, Wo, dst_pitch, s2d, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2])
  if (src_u8) { if (dtype == HRP_F32) HRP_RZ(uint8_t, float); else HRP_RZ(uint8_t, bf16_t); }
  else { if (dtype == HRP_F32) HRP_RZ(float, float); else HRP_RZ(float, bf16_t); }
#undef HRP_RZ
  return check_launch("pil_resize_normalize");
}

extern "C" int hrp_broadcast_hw(const float* src, int src_pitch, void* dst, int dtype, int N, int HW, int C, int dst_pitch, void* stream) {
  HRP_REQUIRE(src && dst && N > 0 && HW > 0 && C > 0 && dst_pitch >= C && src_pitch >= C, "broadcast_hw: bad args");
  dim3 grid((unsigned)(((size_t)HW * C + 255) / 256), N);
  if (dtype == HRP_F32) hipLaunchKernelGGL((broadcast_hw_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, src, src_pitch, (float*)dst, HW, C, dst_pitch);
  else hipLaunchKernelGGL((broadcast_hw_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, src, src_pitch, (bf16_t*)dst, HW, C, dst_pitch);
  return check_launch("broadcast_hw");
}

extern "C" int hrp_bilinear_nhwc_to_nchw(const void* src, int dtype, int N, int h, int w, int C, int pitch, float* dst, int H, int W,
                                         int act, void* stream) {
  HRP_REQUIRE(src && dst && N > 0 && h > 0 && w > 0 && C > 0 && pitch >= C && H > 0 && W > 0, "bilinear: bad args");
  HRP_REQUIRE(act == 0 || act == 1, "bilinear: act must be 0 (none) or 1 (sigmoid)");
  dim3 grid(cdiv(H * W, 256), N);
  if (dtype == HRP_F32) hipLaunchKernelGGL((bilinear_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, h, w, C, pitch, dst, H, W, act);
  else hipLaunchKernelGGL((bilinear_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, h, w, C, pitch, dst, H, W, act);
  return check_launch("bilinear_nhwc_to_nchw");
}
