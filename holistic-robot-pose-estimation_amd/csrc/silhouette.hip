// Soft-silhouette rasteriser of the render-and-compare path (row f-3; reference lib/utils/mesh_renderer.py:78-109 composes
// pytorch3d 0.7.4's MeshRasterizer + SoftSilhouetteShader with BlendParams(sigma = 1e-8), blur_radius = log(1 / 1e-4 - 1) * sigma,
// faces_per_pixel = 100, PerspectiveCameras(focal = (-fx, -fy), in_ndc = False)).
//
// PARITY UNPINNED: pytorch3d is neither in the reference tree nor in the build container and the reference has no fixture of a
// rendered mask.  What is restated here is pytorch3d's published algorithm:
//   per pixel centre c = (x + 0.5, y + 0.5) and face (v0, v1, v2) in screen space (u = fx X / Z + cx: the negative focal lengths
//   undo pytorch3d's left / up NDC axes):
//     inside    all three barycentric edge functions > 0
//     d         squared distance from c to the nearest of the three edge SEGMENTS, in NDC units: (2 / min(H, W))^2 pixels^2
//     kept      inside, or d < blur_radius;   signed s = inside ? -d : d
//     p         sigmoid(-s / sigma)
//   alpha(pixel) = 1 - prod over kept faces of (1 - p)                      (sigmoid_alpha_blend; the trainer takes channel 3)
//   backward: d alpha / d s_k = -prod(1 - p) p_k / sigma, d s / d (edge end points) of the nearest segment, atomically summed per vertex.
// faces_per_pixel = 100 (mesh_renderer.py:99): pytorch3d keeps the 100 faces NEAREST in z of those that pass the test above; here
//   every kept face enters the product, which is the same number wherever at most 100 faces are kept at a pixel - always, for a
//   robot's visual mesh (kept = the pixel CENTRE inside the face or within sqrt(blur_radius) = 0.036 px of it: the depth complexity of
//   the mesh, < 10).  hrp_silhouette_desc.count returns the per-pixel number so that a caller can verify it
//   (URDFRobot.render_silhouette(check_faces_per_pixel=True) raises above 100) instead of assuming it.
// Faces behind the camera: the reference's RasterizationSettings leave z_clip_value = None and its PerspectiveCameras carry no znear
//   (mesh_renderer.py:96-105), so pytorch3d does NOT clip at a near plane; its rasteriser then drops a face when ANY vertex has
//   z < kEpsilon = 1e-8 ("z_invalid = zlims.x < kEpsilon" in rasterize_meshes.cu) - the rule load_face applies.  z ordering is
//   irrelevant for the silhouette.
//
// Face-parallel: a robot's visual mesh projects to triangles of a few pixels at 320 x 240, so one thread per (sample, face)
// walks the face's bounding box.  The per-pixel product is accumulated as a FIXED-POINT sum of log(1 - p) (64-bit integer
// atomics: order independent, bit-reproducible); vertex gradients use float atomics (their order is not fixed).
#include "hrp_common.h"

namespace hrp {

constexpr float S2R_EPS = 1e-8f;
constexpr float S2R_LOG_FLOOR = -100.f;          // log(1 - p) of a pixel strictly inside a face (exp(-100) is 0 in fp32)
constexpr double S2R_FIX = 4294967296.0;         // 2^32

struct Face2D {
  float x0, y0, x1, y1, x2, y2, area;
  int v0, v1, v2;
  int px0, px1, py0, py1;      // pixel range of the (blurred) bounding box, empty when px0 > px1
};

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

// squared distance from p to segment (a, b); tt = clamped parameter of the closest point
__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by, float& tt) {
  const float dx = bx - ax, dy = by - ay;
  const float l2 = dx * dx + dy * dy;
  if (l2 <= S2R_EPS) { tt = 1.f; return (px - bx) * (px - bx) + (py - by) * (py - by); }
  float t = ((px - ax) * dx + (py - ay) * dy) / l2;
  t = fminf(fmaxf(t, 0.f), 1.f);
  tt = t;
  const float qx = ax + t * dx - px, qy = ay + t * dy - py;
  return qx * qx + qy * qy;
}

__device__ __forceinline__ bool load_face(const hrp_silhouette_desc& d, int b, int f, float k2, Face2D& F) {
  F.v0 = d.faces[3 * f]; F.v1 = d.faces[3 * f + 1]; F.v2 = d.faces[3 * f + 2];
  const float* uv = d.uv + (size_t)b * d.V * 2;
  const float* xyz = d.xyz + (size_t)b * d.V * 3;
  if (fminf(fminf(xyz[3 * F.v0 + 2], xyz[3 * F.v1 + 2]), xyz[3 * F.v2 + 2]) < S2R_EPS) return false;      // behind the camera
  F.x0 = uv[2 * F.v0]; F.y0 = uv[2 * F.v0 + 1];
  F.x1 = uv[2 * F.v1]; F.y1 = uv[2 * F.v1 + 1];
  F.x2 = uv[2 * F.v2]; F.y2 = uv[2 * F.v2 + 1];
  F.area = edge_fn(F.x2, F.y2, F.x0, F.y0, F.x1, F.y1);
  if (fabsf(F.area * k2) <= S2R_EPS) return false;                                                          // zero-area face
  const float r = sqrtf(d.blur_radius / k2);                // blur radius in pixels
  const float xmin = fminf(fminf(F.x0, F.x1), F.x2) - r, xmax = fmaxf(fmaxf(F.x0, F.x1), F.x2) + r;
  const float ymin = fminf(fminf(F.y0, F.y1), F.y2) - r, ymax = fmaxf(fmaxf(F.y0, F.y1), F.y2) + r;
  // pixel centres x + 0.5 in [xmin, xmax]
  F.px0 = max(0, (int)ceilf(xmin - 0.5f)); F.px1 = min(d.W - 1, (int)floorf(xmax - 0.5f));
  F.py0 = max(0, (int)ceilf(ymin - 0.5f)); F.py1 = min(d.H - 1, (int)floorf(ymax - 0.5f));
  return F.px0 <= F.px1 && F.py0 <= F.py1;
}

// -> kept?; s = signed squared NDC distance; which edge is nearest (0: v0v1, 1: v1v2, 2: v2v0) and its parameter
__device__ __forceinline__ bool pixel_face(const Face2D& F, float cx, float cy, float k2, float blur, float& s, int& e, float& tt) {
  const float w0 = edge_fn(cx, cy, F.x1, F.y1, F.x2, F.y2) / F.area;
  const float w1 = edge_fn(cx, cy, F.x2, F.y2, F.x0, F.y0) / F.area;
  const float w2 = edge_fn(cx, cy, F.x0, F.y0, F.x1, F.y1) / F.area;
  const bool inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
  float t0, t1, t2;
  const float d0 = seg_dist2(cx, cy, F.x0, F.y0, F.x1, F.y1, t0);
  const float d1 = seg_dist2(cx, cy, F.x1, F.y1, F.x2, F.y2, t1);
  const float d2 = seg_dist2(cx, cy, F.x2, F.y2, F.x0, F.y0, t2);
  float dm = d0; e = 0; tt = t0;
  if (d1 < dm) { dm = d1; e = 1; tt = t1; }
  if (d2 < dm) { dm = d2; e = 2; tt = t2; }
  const float dn = dm * k2;
  if (!inside && dn >= blur) return false;
  s = inside ? -dn : dn;
  return true;
}

// A face whose (blurred) bounding box holds more than S2R_BIG pixel centres is not walked by its own lane: the wave takes such faces
// one after the other, all 64 lanes sharing the box (a robot close to the camera projects single triangles onto thousands of pixels;
// one lane walking them serialises the wave behind it - ADVICE r4 / VERDICT r5 item 7).
constexpr int S2R_BIG = 64;

__device__ __forceinline__ Face2D face_from_lane(const Face2D& F, int src) {
  Face2D G;
  G.x0 = __shfl(F.x0, src, 64); G.y0 = __shfl(F.y0, src, 64); G.x1 = __shfl(F.x1, src, 64); G.y1 = __shfl(F.y1, src, 64);
  G.x2 = __shfl(F.x2, src, 64); G.y2 = __shfl(F.y2, src, 64); G.area = __shfl(F.area, src, 64);
  G.v0 = __shfl(F.v0, src, 64); G.v1 = __shfl(F.v1, src, 64); G.v2 = __shfl(F.v2, src, 64);
  G.px0 = __shfl(F.px0, src, 64); G.px1 = __shfl(F.px1, src, 64); G.py0 = __shfl(F.py0, src, 64); G.py1 = __shfl(F.py1, src, 64);
  return G;
}

__device__ __forceinline__ void fwd_pixel(const hrp_silhouette_desc& d, const Face2D& F, int b, int x, int y, float k2, long long* lp) {
  float s, tt; int e;
  if (!pixel_face(F, x + 0.5f, y + 0.5f, k2, d.blur_radius, s, e, tt)) return;
  // log(1 - sigmoid(-s / sigma)) = -softplus(-s / sigma)
  const float a = -s / d.sigma;
  float l = a > 30.f ? -a : -log1pf(expf(a));
  l = fmaxf(l, S2R_LOG_FLOOR);
  atomicAdd((unsigned long long*)(lp + y * d.W + x), (unsigned long long)(long long)llrint((double)l * S2R_FIX));
  if (d.count) atomicAdd(d.count + (size_t)b * d.H * d.W + y * d.W + x, 1);
}

__global__ __launch_bounds__(256) void silhouette_fwd_kernel(const hrp_silhouette_desc d) {
  const int f = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, lane = threadIdx.x & 63;
  const float sc = 2.f / (float)min(d.H, d.W), k2 = sc * sc;
  Face2D F = {};
  const bool have = f < d.F && load_face(d, b, f, k2, F);
  const int npx = have ? (F.px1 - F.px0 + 1) * (F.py1 - F.py0 + 1) : 0;
  long long* lp = (long long*)d.logp + (size_t)b * d.H * d.W;
  if (have && npx <= S2R_BIG)
    for (int y = F.py0; y <= F.py1; ++y)
      for (int x = F.px0; x <= F.px1; ++x) fwd_pixel(d, F, b, x, y, k2, lp);
  unsigned long long big = __ballot(npx > S2R_BIG);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const Face2D G = face_from_lane(F, src);
    const int w = G.px1 - G.px0 + 1, n = w * (G.py1 - G.py0 + 1);
    for (int i = lane; i < n; i += 64) fwd_pixel(d, G, b, G.px0 + i % w, G.py0 + i / w, k2, lp);
  }
}

__global__ __launch_bounds__(256) void silhouette_alpha_kernel(const hrp_silhouette_desc d) {
  const size_t n = (size_t)d.B * d.H * d.W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    d.alpha[i] = 1.f - expf((float)((double)((const long long*)d.logp)[i] / S2R_FIX));
}

__device__ __forceinline__ bool bwd_pixel(const hrp_silhouette_desc& d, const Face2D& F, int x, int y, float k2, const long long* lp,
                                          const float* ga, float (&g)[3][2]) {
  const float gp = ga[y * d.W + x];
  if (gp == 0.f) return false;
  const float cx = x + 0.5f, cy = y + 0.5f;
  float s, tt; int e;
  if (!pixel_face(F, cx, cy, k2, d.blur_radius, s, e, tt)) return false;
  const float P = expf((float)((double)lp[y * d.W + x] / S2R_FIX));
  if (P == 0.f) return false;
  const float p = 1.f / (1.f + expf(s / d.sigma));
  // d alpha / d s = -P p / sigma;  s = +-k2 * (pixel distance)^2
  const float gd = gp * (-P * p / d.sigma) * (s < 0.f ? -k2 : k2);
  const float ax = e == 0 ? F.x0 : e == 1 ? F.x1 : F.x2, ay = e == 0 ? F.y0 : e == 1 ? F.y1 : F.y2;
  const float bx = e == 0 ? F.x1 : e == 1 ? F.x2 : F.x0, by = e == 0 ? F.y1 : e == 1 ? F.y2 : F.y0;
  const float rx = cx - (ax + tt * (bx - ax)), ry = cy - (ay + tt * (by - ay));
  const bool degenerate = (bx - ax) * (bx - ax) + (by - ay) * (by - ay) <= S2R_EPS;
  const float wa = degenerate ? 0.f : -2.f * (1.f - tt), wb = -2.f * tt;      // d (dist^2) / d a = wa r, / d b = wb r
  const int ia = e, ib = (e + 1) % 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {       // (static indices: g stays in registers)
    if (k == ia) { g[k][0] += gd * wa * rx; g[k][1] += gd * wa * ry; }
    if (k == ib) { g[k][0] += gd * wb * rx; g[k][1] += gd * wb * ry; }
  }
  return true;
}

__global__ __launch_bounds__(256) void silhouette_bwd_kernel(const hrp_silhouette_desc d, const float* __restrict__ d_alpha, float* __restrict__ d_uv) {
  const int f = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, lane = threadIdx.x & 63;
  const float sc = 2.f / (float)min(d.H, d.W), k2 = sc * sc;
  Face2D F = {};
  const bool have = f < d.F && load_face(d, b, f, k2, F);
  const int npx = have ? (F.px1 - F.px0 + 1) * (F.py1 - F.py0 + 1) : 0;
  const long long* lp = (const long long*)d.logp + (size_t)b * d.H * d.W;
  const float* ga = d_alpha + (size_t)b * d.H * d.W;
  float* o = d_uv + (size_t)b * d.V * 2;
  if (have && npx <= S2R_BIG) {
    float g[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    bool any = false;
    for (int y = F.py0; y <= F.py1; ++y)
      for (int x = F.px0; x <= F.px1; ++x) any |= bwd_pixel(d, F, x, y, k2, lp, ga, g);
    if (any) {
      const int vid[3] = {F.v0, F.v1, F.v2};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (g[k][0] != 0.f) atomicAdd(o + 2 * vid[k], g[k][0]);
        if (g[k][1] != 0.f) atomicAdd(o + 2 * vid[k] + 1, g[k][1]);
      }
    }
  }
  unsigned long long big = __ballot(npx > S2R_BIG);
  while (big) {            // large faces: the wave shares the box, lane sums are folded across the wave, one lane adds them
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const Face2D G = face_from_lane(F, src);
    const int w = G.px1 - G.px0 + 1, n = w * (G.py1 - G.py0 + 1);
    float g[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    for (int i = lane; i < n; i += 64) bwd_pixel(d, G, G.px0 + i % w, G.py0 + i / w, k2, lp, ga, g);
    const int vid[3] = {G.v0, G.v1, G.v2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float gx = wave_sum(g[k][0]), gy = wave_sum(g[k][1]);
      if (lane == 0) {
        if (gx != 0.f) atomicAdd(o + 2 * vid[k], gx);
        if (gy != 0.f) atomicAdd(o + 2 * vid[k] + 1, gy);
      }
    }
  }
}

}  // namespace hrp

using namespace hrp;

static int silhouette_check(const hrp_silhouette_desc* d) {
  HRP_REQUIRE(d && d->uv && d->xyz && d->faces && d->alpha && d->logp, "silhouette: null pointer");
  HRP_REQUIRE(d->B > 0 && d->V > 0 && d->F > 0 && d->H > 0 && d->W > 0 && d->sigma > 0.f && d->blur_radius >= 0.f,
              "silhouette: B=%d V=%d F=%d H=%d W=%d sigma=%g blur=%g", d->B, d->V, d->F, d->H, d->W, d->sigma, d->blur_radius);
  return HRP_OK;
}

extern "C" int hrp_silhouette_fwd(const hrp_silhouette_desc* d, void* stream) {
  int rc = silhouette_check(d);
  if (rc != HRP_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  zero_async(d->logp, (size_t)d->B * d->H * d->W * 8, s);
  if (d->count) zero_async(d->count, (size_t)d->B * d->H * d->W * 4, s);
  hipLaunchKernelGGL(silhouette_fwd_kernel, dim3(cdiv(d->F, 256), d->B), dim3(256), 0, s, *d);
  const size_t n = (size_t)d->B * d->H * d->W;
  hipLaunchKernelGGL(silhouette_alpha_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s, *d);
  return check_launch("silhouette_fwd");
}

extern "C" int hrp_silhouette_bwd(const hrp_silhouette_desc* d, const float* d_alpha, float* d_uv, void* stream) {
  int rc = silhouette_check(d);
  if (rc != HRP_OK) return rc;
  HRP_REQUIRE(d_alpha && d_uv, "silhouette_bwd: null gradient pointer");
  hipStream_t s = (hipStream_t)stream;
  zero_async(d_uv, (size_t)d->B * d->V * 2 * 4, s);
  hipLaunchKernelGGL(silhouette_bwd_kernel, dim3(cdiv(d->F, 256), d->B), dim3(256), 0, s, *d, d_alpha, d_uv);
  return check_launch("silhouette_bwd");
}
