// Pose heads: one-pass 3-D soft-argmax, camera geometry, forward kinematics + projection.
#include "hrp_common.h"

namespace hrp {

// =================================================================================================
// 3-D soft-argmax (reference lib/utils/integral.py:147-177): per (sample, joint) softmax over D*H*W
// logits, then E[w], E[h], E[d].  The reference makes >= 5 passes over the [B, J*D, H, W] tensor;
// here one workgroup streams the joint's logits ONCE with an online softmax carrying the three
// weighted sums.  NHWC logits: the D depth bins of joint j are D contiguous channels of each pixel.
// =================================================================================================
struct SmState {
  float m, s, sx, sy, sz;
};
__device__ __forceinline__ void sm_merge(SmState& a, const SmState& b) {
  float m = fmaxf(a.m, b.m);
  float fa = __expf(a.m - m), fb = __expf(b.m - m);
  a.s = a.s * fa + b.s * fb;
  a.sx = a.sx * fa + b.sx * fb;
  a.sy = a.sy * fa + b.sy * fb;
  a.sz = a.sz * fa + b.sz * fb;
  a.m = m;
}

template <typename T>
__global__ __launch_bounds__(256) void softargmax_fwd_kernel(const void* __restrict__ logits, int J, int D, int H, int W, int pitch,
                                                             int root, int fix_root, float* __restrict__ uvd, float* __restrict__ ms) {
  constexpr int VEC = Elem<T>::VEC;
  const int j = blockIdx.x, b = blockIdx.y;
  const int nv = D / VEC;          // vectors per pixel for this joint
  const int ppi = 256 / nv;        // pixels per iteration
  const int vi = threadIdx.x % nv, pl = threadIdx.x / nv;
  SmState st{-INFINITY, 0.f, 0.f, 0.f, 0.f};
  const int HW = H * W;
  if (pl < ppi) {
    for (int p = pl; p < HW; p += ppi) {
      const int y = p / W, x = p - y * W;
      float f[VEC];
      Elem<T>::unpack(*(const uint4*)((const char*)logits + (((size_t)b * HW + p) * pitch + j * D + vi * VEC) * Elem<T>::SZ), f);
      float mx = f[0];
#pragma unroll
      for (int i = 1; i < VEC; ++i) mx = fmaxf(mx, f[i]);
      if (mx > st.m) {
        float sc = __expf(st.m - mx);  // exp(-inf) = 0 on the first element
        st.s *= sc; st.sx *= sc; st.sy *= sc; st.sz *= sc;
        st.m = mx;
      }
      float es = 0.f, ez = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float e = __expf(f[i] - st.m);
        es += e;
        ez += e * (float)(vi * VEC + i);
      }
      st.s += es; st.sx += es * (float)x; st.sy += es * (float)y; st.sz += ez;
    }
  }
  // wave reduce then block reduce
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    SmState ot;
    ot.m = __shfl_xor(st.m, o, 64); ot.s = __shfl_xor(st.s, o, 64);
    ot.sx = __shfl_xor(st.sx, o, 64); ot.sy = __shfl_xor(st.sy, o, 64); ot.sz = __shfl_xor(st.sz, o, 64);
    if (ot.m > -INFINITY || st.m > -INFINITY) {
      if (st.m == -INFINITY) st = ot;
      else if (ot.m > -INFINITY) sm_merge(st, ot);
    }
  }
  __shared__ SmState part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = st;
  __syncthreads();
  if (threadIdx.x == 0) {
    SmState a = part[0];
    for (int w = 1; w < 4; ++w) {
      if (part[w].m == -INFINITY) continue;
      if (a.m == -INFINITY) a = part[w];
      else sm_merge(a, part[w]);
    }
    float inv = 1.f / a.s;
    float* o = uvd + ((size_t)b * J + j) * 3;
    o[0] = a.sx * inv / (float)W - 0.5f;
    o[1] = a.sy * inv / (float)H - 0.5f;
    o[2] = (fix_root && j == root) ? 0.f : a.sz * inv / (float)D - 0.5f;
    ms[((size_t)b * J + j) * 2 + 0] = a.m;
    ms[((size_t)b * J + j) * 2 + 1] = a.s;
    // expected depth bin kept for the backward of the non-root joints only; root (fix_root) has no d-gradient
  }
}

// d logit = p * ( gu (x - E[x]) / W + gv (y - E[y]) / H + gd (d - E[d]) / D )
template <typename T>
__global__ __launch_bounds__(256) void softargmax_bwd_kernel(const void* __restrict__ logits, int J, int D, int H, int W, int pitch,
                                                             int root, int fix_root, const float* __restrict__ uvd,
                                                             const float* __restrict__ ms, const float* __restrict__ duvd,
                                                             void* __restrict__ dlogits, int dpitch) {
  constexpr int VEC = Elem<T>::VEC;
  const int b = blockIdx.y;
  const int C = J * D, nv = C / VEC;
  const long total = (long)H * W * nv;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long)gridDim.x * 256) {
    const int cv = v % nv;
    const int p = v / nv;
    const int y = p / W, x = p - y * W;
    const int c = cv * VEC, j = c / D, d0 = c - j * D;
    const float* g = duvd + ((size_t)b * J + j) * 3;
    const float* o = uvd + ((size_t)b * J + j) * 3;
    const float m = ms[((size_t)b * J + j) * 2], inv = 1.f / ms[((size_t)b * J + j) * 2 + 1];
    const bool nod = fix_root && j == root;
    const float gu = g[0] / (float)W, gv = g[1] / (float)H, gd = nod ? 0.f : g[2] / (float)D;
    const float ex = (o[0] + 0.5f) * (float)W, ey = (o[1] + 0.5f) * (float)H, ed = (o[2] + 0.5f) * (float)D;
    const float base = gu * ((float)x - ex) + gv * ((float)y - ey);
    float f[VEC], r[VEC];
    Elem<T>::unpack(*(const uint4*)((const char*)logits + (((size_t)b * H * W + p) * pitch + c) * Elem<T>::SZ), f);
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = __expf(f[i] - m) * inv * (base + gd * ((float)(d0 + i) - ed));
    *(uint4*)((char*)dlogits + (((size_t)b * H * W + p) * dpitch + c) * Elem<T>::SZ) = Elem<T>::pack(r);
  }
}

// =================================================================================================
// Camera geometry of the heads (reference full_net.py:281-305, transforms.py:33-73, 133-162).
// One thread per sample.
// =================================================================================================
struct InvK {
  float a, b, c, d;  // [[a 0 b],[0 c d],[0 0 1]]
};
__device__ __forceinline__ InvK inv_intrinsics(const float* K) {
  // fp64 divide stored to fp32, as transforms.py:150-154
  double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  InvK r;
  r.a = (float)(1.0 / fx); r.b = (float)(-cx / fx);
  r.c = (float)(1.0 / fy); r.d = (float)(-cy / fy);
  return r;
}

__global__ void pose_geometry_fwd_kernel(const float* __restrict__ gamma, const float* __restrict__ kval, const float* __restrict__ uvd,
                                         const float* __restrict__ K, int B, int J, int root, float S, float DF,
                                         float* __restrict__ depth, float* __restrict__ xyz, float* __restrict__ root_uv,
                                         float* __restrict__ trans) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  InvK ik = inv_intrinsics(K + 9 * b);
  float z = gamma[b] * kval[b] / 1000.0f;
  depth[b] = z;
  for (int j = 0; j < J; ++j) {
    const float* o = uvd + ((size_t)b * J + j) * 3;
    float U = (o[0] + 0.5f) * S, V = (o[1] + 0.5f) * S;
    float za = o[2] * DF + z;
    float* x = xyz + ((size_t)b * J + j) * 3;
    x[0] = (ik.a * U + ik.b) * za;
    x[1] = (ik.c * V + ik.d) * za;
    x[2] = za;
  }
  const float* o = uvd + ((size_t)b * J + root) * 3;
  float ru = (o[0] + 0.5f) * S, rv = (o[1] + 0.5f) * S;
  root_uv[2 * b] = ru; root_uv[2 * b + 1] = rv;
  trans[3 * b] = ik.a * (ru * z) + ik.b * z;
  trans[3 * b + 1] = ik.c * (rv * z) + ik.d * z;
  trans[3 * b + 2] = z;
}

__global__ void pose_geometry_bwd_kernel(const float* __restrict__ gamma, const float* __restrict__ kval, const float* __restrict__ uvd,
                                         const float* __restrict__ K, int B, int J, int root, int fix_root, float S, float DF,
                                         const float* __restrict__ d_depth, const float* __restrict__ d_xyz,
                                         const float* __restrict__ d_root_uv, const float* __restrict__ d_trans,
                                         float* __restrict__ d_gamma, float* __restrict__ d_uvd) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  InvK ik = inv_intrinsics(K + 9 * b);
  float z = gamma[b] * kval[b] / 1000.0f;
  float dz = d_depth ? d_depth[b] : 0.f;
  for (int j = 0; j < J; ++j) {
    const float* o = uvd + ((size_t)b * J + j) * 3;
    float U = (o[0] + 0.5f) * S, V = (o[1] + 0.5f) * S;
    float za = o[2] * DF + z;
    float r0 = ik.a * U + ik.b, r1 = ik.c * V + ik.d;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (d_xyz) { const float* g = d_xyz + ((size_t)b * J + j) * 3; g0 = g[0]; g1 = g[1]; g2 = g[2]; }
    float dza = g0 * r0 + g1 * r1 + g2;
    float* du = d_uvd + ((size_t)b * J + j) * 3;
    du[0] = g0 * za * ik.a * S;
    du[1] = g1 * za * ik.c * S;
    du[2] = (fix_root && j == root) ? 0.f : dza * DF;
    dz += dza;
  }
  const float* o = uvd + ((size_t)b * J + root) * 3;
  float ru = (o[0] + 0.5f) * S, rv = (o[1] + 0.5f) * S;
  float t0 = d_trans ? d_trans[3 * b] : 0.f, t1 = d_trans ? d_trans[3 * b + 1] : 0.f, t2 = d_trans ? d_trans[3 * b + 2] : 0.f;
  float dru = t0 * ik.a * z + (d_root_uv ? d_root_uv[2 * b] : 0.f);
  float drv = t1 * ik.c * z + (d_root_uv ? d_root_uv[2 * b + 1] : 0.f);
  dz += t0 * (ik.a * ru + ik.b) + t1 * (ik.c * rv + ik.d) + t2;
  float* du = d_uvd + ((size_t)b * J + root) * 3;
  du[0] += dru * S;
  du[1] += drv * S;
  d_gamma[b] = dz * kval[b] / 1000.0f;
}

// =================================================================================================
// Forward kinematics + projection, one wavefront per sample.
//   forward : lane k computes keypoint k (walks its own base->link chain)
//   backward: lane l carries the tangent of input parameter l (q_0..q_dof-1, rot6d_0..5, t_0..2)
//             through the SAME templated code with dual numbers (forward-mode AD, exact), then dots
//             its tangent outputs with the incoming gradients: one lane = one gradient entry.
// Arithmetic follows lib/utils/urdfpytorch/urdf.py:2380-2390, 2427-2462, 3115-3140 (T = T_parent * origin *
// Rot(axis, q), fp32), geometries.py:100-115 (rot6d), urdf_robot.py:169-199 (re-rooting; closed-form rigid
// inverse instead of torch.linalg.inv) and transforms.py:7-21.
// =================================================================================================
struct Dual {
  float v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { float q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator+(Dual a, float b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual dsqrt(Dual a) { float s = sqrtf(a.v); return {s, a.d / (2.f * s)}; }
__device__ __forceinline__ float dsqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ Dual dsin(Dual a) { return {sinf(a.v), cosf(a.v) * a.d}; }
__device__ __forceinline__ Dual dcos(Dual a) { return {cosf(a.v), -sinf(a.v) * a.d}; }
__device__ __forceinline__ float dsin(float a) { return sinf(a); }
__device__ __forceinline__ float dcos(float a) { return cosf(a); }
template <class S> __device__ __forceinline__ S lift(float v);
template <> __device__ __forceinline__ float lift<float>(float v) { return v; }
template <> __device__ __forceinline__ Dual lift<Dual>(float v) { return {v, 0.f}; }

template <class S>
struct Rigid {  // 3x4 [R | t]
  S r[3][3], t[3];
};

template <class S>
__device__ __forceinline__ Rigid<S> rigid_identity() {
  Rigid<S> T;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) T.r[i][j] = lift<S>(i == j ? 1.f : 0.f);
    T.t[i] = lift<S>(0.f);
  }
  return T;
}

template <class S>
__device__ __forceinline__ Rigid<S> rigid_mul(const Rigid<S>& A, const Rigid<S>& B) {
  Rigid<S> C;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) C.r[i][j] = A.r[i][0] * B.r[0][j] + A.r[i][1] * B.r[1][j] + A.r[i][2] * B.r[2][j];
    C.t[i] = A.r[i][0] * B.t[0] + A.r[i][1] * B.t[1] + A.r[i][2] * B.t[2] + A.t[i];
  }
  return C;
}

template <class S>
__device__ __forceinline__ Rigid<S> rigid_inverse(const Rigid<S>& A) {
  Rigid<S> C;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) C.r[i][j] = A.r[j][i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) C.t[i] = lift<S>(0.f) - (C.r[i][0] * A.t[0] + C.r[i][1] * A.t[1] + C.r[i][2] * A.t[2]);
  return C;
}

// child pose of joint j relative to its parent: origin * motion(q)
template <class S>
__device__ __forceinline__ Rigid<S> joint_pose(const hrp_fk_chain* ch, int j, const S* q) {
  Rigid<S> O;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int k = 0; k < 3; ++k) O.r[i][k] = lift<S>(ch->origin[j][i * 4 + k]);
    O.t[i] = lift<S>(ch->origin[j][i * 4 + 3]);
  }
  const int type = ch->type[j], cfg = ch->cfg[j];
  if (type == 0 || cfg < 0) return O;
  S qv = ch->mimic_mul[j] * q[cfg] + ch->mimic_off[j];
  const float ax = ch->axis[j][0], ay = ch->axis[j][1], az = ch->axis[j][2];
  Rigid<S> M = rigid_identity<S>();
  if (type == 1) {
    S s = dsin(qv), c = dcos(qv);
    S omc = 1.f - c;
    const float a[3] = {ax, ay, az};
    const float sk[3][3] = {{0.f, -az, ay}, {az, 0.f, -ax}, {-ay, ax, 0.f}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        S v = omc * (a[i] * a[k]) + s * sk[i][k];
        if (i == k) v = c + v;
        M.r[i][k] = v;
      }
  } else {
    M.t[0] = qv * ax; M.t[1] = qv * ay; M.t[2] = qv * az;
  }
  return rigid_mul(O, M);
}

// pose of the child frame of joint `leaf` in the base frame (leaf = -1: identity)
template <class S>
__device__ __forceinline__ Rigid<S> frame_pose(const hrp_fk_chain* ch, int leaf, const S* q) {
  Rigid<S> T = rigid_identity<S>();
  unsigned mask = 0;
  for (int j = leaf; j >= 0; j = ch->parent[j]) mask |= 1u << j;
  for (int j = 0; j < ch->njoints; ++j)
    if (mask & (1u << j)) T = rigid_mul(T, joint_pose<S>(ch, j, q));
  return T;
}

template <class S>
__device__ __forceinline__ Rigid<S> base_to_cam(const S* r6, const S* tr, const int rd = 6) {
  if (rd == 4) {
    // quaternion (w, x, y, z), geometries.py:21-41: normalised by (norm + 1e-9)
    const S n = dsqrt(r6[0] * r6[0] + r6[1] * r6[1] + r6[2] * r6[2] + r6[3] * r6[3]) + lift<S>(1e-9f);
    const S w = r6[0] / n, x = r6[1] / n, y = r6[2] / n, z = r6[3] / n;
    const S w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
    const S wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
    const S two = lift<S>(2.f);
    Rigid<S> T;
    T.r[0][0] = w2 + x2 - y2 - z2; T.r[0][1] = two * xy - two * wz; T.r[0][2] = two * wy + two * xz;
    T.r[1][0] = two * wz + two * xy; T.r[1][1] = w2 - x2 + y2 - z2; T.r[1][2] = two * yz - two * wx;
    T.r[2][0] = two * xz - two * wy; T.r[2][1] = two * wx + two * yz; T.r[2][2] = w2 - x2 - y2 + z2;
    T.t[0] = tr[0]; T.t[1] = tr[1]; T.t[2] = tr[2];
    return T;
  }
  // geometries.py:100-115
  S ax = r6[0], ay = r6[1], az = r6[2], bx = r6[3], by = r6[4], bz = r6[5];
  S n = dsqrt(ax * ax + ay * ay + az * az);
  S x0 = ax / n, x1 = ay / n, x2 = az / n;
  S z0 = x1 * bz - x2 * by, z1 = x2 * bx - x0 * bz, z2 = x0 * by - x1 * bx;
  S zn = dsqrt(z0 * z0 + z1 * z1 + z2 * z2);
  z0 = z0 / zn; z1 = z1 / zn; z2 = z2 / zn;
  S y0 = z1 * x2 - z2 * x1, y1 = z2 * x0 - z0 * x2, y2 = z0 * x1 - z1 * x0;
  Rigid<S> T;
  T.r[0][0] = x0; T.r[0][1] = x1; T.r[0][2] = x2;
  T.r[1][0] = y0; T.r[1][1] = y1; T.r[1][2] = y2;
  T.r[2][0] = z0; T.r[2][1] = z1; T.r[2][2] = z2;
  T.t[0] = tr[0]; T.t[1] = tr[1]; T.t[2] = tr[2];
  return T;
}

// M = B2C (root == 0) or B2C * T_root^-1
template <class S>
__device__ __forceinline__ Rigid<S> camera_from_base(const hrp_fk_chain* ch, const S* q, const S* r6, const S* tr, int root, const int rd = 6) {
  Rigid<S> M = base_to_cam<S>(r6, tr, rd);
  if (root > 0) M = rigid_mul(M, rigid_inverse(frame_pose<S>(ch, ch->kp_frame[root], q)));
  return M;
}

template <class S>
__device__ __forceinline__ void keypoint(const hrp_fk_chain* ch, const Rigid<S>& M, const S* q, int k, S* p) {
  Rigid<S> P = rigid_mul(M, frame_pose<S>(ch, ch->kp_frame[k], q));
  const float ox = ch->kp_offset[k][0], oy = ch->kp_offset[k][1], oz = ch->kp_offset[k][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = P.r[i][0] * ox + P.r[i][1] * oy + P.r[i][2] * oz + P.t[i];
}

template <class S>
__device__ __forceinline__ void project(const float* K, const S* p, S* uv) {
  S h0 = K[0] * p[0] + K[1] * p[1] + K[2] * p[2];
  S h1 = K[3] * p[0] + K[4] * p[1] + K[5] * p[2];
  S h2 = K[6] * p[0] + K[7] * p[1] + K[8] * p[2];
  uv[0] = h0 / h2; uv[1] = h1 / h2;
}

__global__ __launch_bounds__(64) void fk_project_fwd_kernel(const hrp_fk_chain* __restrict__ ch, const float* __restrict__ q,
                                                            const float* __restrict__ r6, const float* __restrict__ tr,
                                                            const float* __restrict__ K, int root, float* __restrict__ xyz,
                                                            float* __restrict__ uv, float* __restrict__ root_rot, const int rd) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int dof = ch->dof, nkp = ch->nkp;
  float ql[HRP_FK_MAX_JOINTS];
  for (int i = 0; i < dof; ++i) ql[i] = q[(size_t)b * dof + i];
  float rl[6], tl[3];
  for (int i = 0; i < 6; ++i) rl[i] = i < rd ? r6[rd * b + i] : 0.f;
  for (int i = 0; i < 3; ++i) tl[i] = tr[3 * b + i];
  if (lane < nkp) {
    Rigid<float> M = camera_from_base<float>(ch, ql, rl, tl, root, rd);
    float p[3];
    keypoint<float>(ch, M, ql, lane, p);
    float* o = xyz + ((size_t)b * nkp + lane) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    if (K && uv) {
      float w[2];
      project<float>(K + 9 * b, p, w);
      uv[((size_t)b * nkp + lane) * 2] = w[0];
      uv[((size_t)b * nkp + lane) * 2 + 1] = w[1];
    }
  }
  if (root_rot && lane == 63) {
    // urdf_robot.py:113-138: first two rows of (B2C * T_root).R, or its quaternion (geometries.py:63-82)
    Rigid<float> P = base_to_cam<float>(rl, tl, rd);
    if (root > 0) P = rigid_mul(P, frame_pose<float>(ch, ch->kp_frame[root], ql));
    if (rd == 4) {
      float w = sqrtf(fmaxf(1.f + P.r[0][0] + P.r[1][1] + P.r[2][2], 0.f)) / 2.f;
      w = fmaxf(w, 1e-8f);
      const float w4 = 4.f * w;
      float qv[4] = {w, (P.r[2][1] - P.r[1][2]) / w4, (P.r[0][2] - P.r[2][0]) / w4, (P.r[1][0] - P.r[0][1]) / w4};
      const float mag = fmaxf(sqrtf(qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2] + qv[3] * qv[3]), 1e-8f);
      for (int i = 0; i < 4; ++i) root_rot[4 * b + i] = qv[i] / mag;
    } else {
      for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 3; ++k) root_rot[6 * b + i * 3 + k] = P.r[i][k];
    }
  }
}

__global__ __launch_bounds__(64) void fk_project_bwd_kernel(const hrp_fk_chain* __restrict__ ch, const float* __restrict__ q,
                                                            const float* __restrict__ r6, const float* __restrict__ tr,
                                                            const float* __restrict__ K, int root, const float* __restrict__ d_xyz,
                                                            const float* __restrict__ d_uv, float* __restrict__ d_q,
                                                            float* __restrict__ d_r6, float* __restrict__ d_tr, const int rd) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int dof = ch->dof, nkp = ch->nkp;
  const int npar = dof + rd + 3;
  if (lane >= npar) return;
  Dual ql[HRP_FK_MAX_JOINTS], rl[6], tl[3];
  for (int i = 0; i < dof; ++i) ql[i] = {q[(size_t)b * dof + i], lane == i ? 1.f : 0.f};
  for (int i = 0; i < 6; ++i) rl[i] = {i < rd ? r6[rd * b + i] : 0.f, lane == dof + i && i < rd ? 1.f : 0.f};
  for (int i = 0; i < 3; ++i) tl[i] = {tr[3 * b + i], lane == dof + rd + i ? 1.f : 0.f};
  Rigid<Dual> M = camera_from_base<Dual>(ch, ql, rl, tl, root, rd);
  float g = 0.f;
  for (int k = 0; k < nkp; ++k) {
    Dual p[3];
    keypoint<Dual>(ch, M, ql, k, p);
    if (d_xyz) {
      const float* gx = d_xyz + ((size_t)b * nkp + k) * 3;
      g += gx[0] * p[0].d + gx[1] * p[1].d + gx[2] * p[2].d;
    }
    if (d_uv && K) {
      Dual w[2];
      project<Dual>(K + 9 * b, p, w);
      const float* gu = d_uv + ((size_t)b * nkp + k) * 2;
      g += gu[0] * w[0].d + gu[1] * w[1].d;
    }
  }
  if (lane < dof) d_q[(size_t)b * dof + lane] = g;
  else if (lane < dof + rd) d_r6[rd * b + lane - dof] = g;
  else d_tr[3 * b + lane - dof - rd] = g;
}

// standalone pinhole projection (transforms.py:17-21): one thread per point
__global__ void project_fwd_kernel(const float* __restrict__ K, const float* __restrict__ pts, int B, int P, float* __restrict__ uv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * P) return;
  float w[2];
  project<float>(K + 9 * (i / P), pts + 3 * (size_t)i, w);
  uv[2 * (size_t)i] = w[0]; uv[2 * (size_t)i + 1] = w[1];
}
__global__ void project_bwd_kernel(const float* __restrict__ K, const float* __restrict__ pts, const float* __restrict__ duv, int B, int P,
                                   float* __restrict__ dpts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * P) return;
  const float* p = pts + 3 * (size_t)i;
  for (int c = 0; c < 3; ++c) {
    Dual d[3] = {{p[0], c == 0 ? 1.f : 0.f}, {p[1], c == 1 ? 1.f : 0.f}, {p[2], c == 2 ? 1.f : 0.f}};
    Dual w[2];
    project<Dual>(K + 9 * (i / P), d, w);
    dpts[3 * (size_t)i + c] = duv[2 * (size_t)i] * w[0].d + duv[2 * (size_t)i + 1] * w[1].d;
  }
}

// =================================================================================================
// The training loss of configs/panda/full.yaml in ONE launch, with its analytic gradient
// (reference lib/core/function.py:191-322: ten terms assembled from ~60 tensor expressions and two
// per-sample projection loops, :119-122).  One workgroup; a thread walks samples b, b + 256, ...
//   t0 loss_joint   mean (pose - gt)^2                        t5 loss_error3d      mean_j ||xyz_fk - kp3d||
//   t1 loss_rot     mean (rot6d - gt)^2                       t6 loss_error2d      sum m ||uv_fk - kp2d|| / S / #m
//   t2 loss_uv      sum m_root ||root_uv - gt|| / S / #m_root t7 loss_error2d_int  same with uv_int
//   t3 loss_depth   mean |depth - gt_z|                       t8 loss_error3d_int  mean_j ||xyz_int - kp3d||
//   t4 loss_trans   mean e c, e = ||trans - gt||,             t9 loss_error3d_align mean_j ||xyz_fk - xyz_int||
//                   c = exp(-20 e) (no gradient) when mean e > 0.5 else 1        (function.py:245-251)
// uv = (K p)[:2] / (K p)[2] (transforms.py:17-21).  out[0..9] = the terms, out[10] = sum_k w[k] t_k.  Gradients of
// out[10] with respect to every prediction are written to the d_* buffers (norm at 0: gradient 0, as torch.norm).
// =================================================================================================
struct LossArgs {
  const float *pose, *rot, *trans, *root_uv, *depth, *xyz_int, *xyz_fk;                    // predictions
  const float *g_pose, *g_rot, *g_trans, *g_root_uv, *g_kp3d, *g_kp2d, *mask, *K;           // ground truth, intrinsics
  float *d_pose, *d_rot, *d_trans, *d_root_uv, *d_depth, *d_xyz_int, *d_xyz_fk;            // gradients (may be NULL: none)
  float* out;                                                                             // [11]
  float w[10];
  int B, P, J, root, R;
  float S;
};

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void pose_loss_kernel(const LossArgs a) {
  __shared__ float red[4];
  const int B = a.B, P = a.P, J = a.J;
  const float invS = 1.0f / a.S;
  // ---- pass 1: everything the normalisations need ------------------------------------------------------------
  float s_e = 0.f, n_root = 0.f, n_valid = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float dx = a.trans[3 * b] - a.g_trans[3 * b], dy = a.trans[3 * b + 1] - a.g_trans[3 * b + 1],
                dz = a.trans[3 * b + 2] - a.g_trans[3 * b + 2];
    s_e += sqrtf(dx * dx + dy * dy + dz * dz);
    n_root += a.mask[b * J + a.root] != 0.f ? 1.f : 0.f;
    for (int j = 0; j < J; ++j) n_valid += a.mask[b * J + j] != 0.f ? 1.f : 0.f;
  }
  const float mean_e = block_sum(s_e, red) / B;
  n_root = block_sum(n_root, red);
  n_valid = block_sum(n_valid, red);
  const bool damp = mean_e > 0.5f;
  // ---- pass 2: terms and gradients ---------------------------------------------------------------------------
  float t[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) t[k] = 0.f;
  const bool grad = a.d_pose != nullptr;
  for (int b = threadIdx.x; b < B; b += 256) {
    for (int k = 0; k < P; ++k) {
      const float d = a.pose[b * P + k] - a.g_pose[b * P + k];
      t[0] += d * d;
      if (grad) a.d_pose[b * P + k] = a.w[0] * 2.f * d / (B * P);
    }
    for (int k = 0; k < a.R; ++k) {
      const float d = a.rot[b * a.R + k] - a.g_rot[b * a.R + k];
      t[1] += d * d;
      if (grad) a.d_rot[b * a.R + k] = a.w[1] * 2.f * d / (B * a.R);
    }
    {
      const float m = a.mask[b * J + a.root];
      const float ux = (a.root_uv[2 * b] - a.g_root_uv[2 * b]) * invS, uy = (a.root_uv[2 * b + 1] - a.g_root_uv[2 * b + 1]) * invS;
      const float n = sqrtf(ux * ux + uy * uy);
      t[2] += n * m;
      if (grad) {
        const float c = n > 0.f ? a.w[2] * m * invS / (n * n_root) : 0.f;
        a.d_root_uv[2 * b] = c * ux; a.d_root_uv[2 * b + 1] = c * uy;
      }
    }
    {
      const float d = a.depth[b] - a.g_trans[3 * b + 2];
      t[3] += fabsf(d);
      if (grad) a.d_depth[b] = a.w[3] * (d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f) / B;
    }
    {
      const float dx = a.trans[3 * b] - a.g_trans[3 * b], dy = a.trans[3 * b + 1] - a.g_trans[3 * b + 1],
                  dz = a.trans[3 * b + 2] - a.g_trans[3 * b + 2];
      const float e = sqrtf(dx * dx + dy * dy + dz * dz);
      const float c = damp ? expf(-20.0f * e) : 1.f;
      t[4] += e * c;
      if (grad) {
        const float g = e > 0.f ? a.w[4] * c / (e * B) : 0.f;
        a.d_trans[3 * b] = g * dx; a.d_trans[3 * b + 1] = g * dy; a.d_trans[3 * b + 2] = g * dz;
      }
    }
    const float* Kb = a.K + 9 * b;
    for (int j = 0; j < J; ++j) {
      const int o = (b * J + j) * 3;
      const float m = a.mask[b * J + j];
      const float gx = a.g_kp3d[o], gy = a.g_kp3d[o + 1], gz = a.g_kp3d[o + 2];
      const float g2x = a.g_kp2d[(b * J + j) * 2] * invS, g2y = a.g_kp2d[(b * J + j) * 2 + 1] * invS;
      float dfk[3] = {0.f, 0.f, 0.f}, dint[3] = {0.f, 0.f, 0.f};
      const float* pts[2] = {a.xyz_fk + o, a.xyz_int + o};
      float* dps[2] = {dfk, dint};
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float* p = pts[which];
        float* dp = dps[which];
        // 3-D error
        const float ex = p[0] - gx, ey = p[1] - gy, ez = p[2] - gz;
        const float n3 = sqrtf(ex * ex + ey * ey + ez * ez);
        t[which == 0 ? 5 : 8] += n3;
        const float w3 = a.w[which == 0 ? 6 : 8];     // weights: see the order at the launch (kp3d, kp3d_int)
        if (n3 > 0.f) { const float c = w3 / (n3 * B * J); dp[0] += c * ex; dp[1] += c * ey; dp[2] += c * ez; }
        // projected 2-D error
        const float h0 = Kb[0] * p[0] + Kb[1] * p[1] + Kb[2] * p[2], h1 = Kb[3] * p[0] + Kb[4] * p[1] + Kb[5] * p[2],
                    h2 = Kb[6] * p[0] + Kb[7] * p[1] + Kb[8] * p[2];
        const float u = h0 / h2, v = h1 / h2;
        const float qx = u * invS - g2x, qy = v * invS - g2y;
        const float n2 = sqrtf(qx * qx + qy * qy);
        t[which == 0 ? 6 : 7] += n2 * m;
        const float w2 = a.w[which == 0 ? 5 : 7];     // kp2d, kp2d_int
        if (n2 > 0.f && m != 0.f) {
          const float c = w2 * m * invS / (n2 * n_valid);
          const float du = c * qx, dv = c * qy;
          const float dh0 = du / h2, dh1 = dv / h2, dh2 = -(du * h0 + dv * h1) / (h2 * h2);
          dp[0] += Kb[0] * dh0 + Kb[3] * dh1 + Kb[6] * dh2;
          dp[1] += Kb[1] * dh0 + Kb[4] * dh1 + Kb[7] * dh2;
          dp[2] += Kb[2] * dh0 + Kb[5] * dh1 + Kb[8] * dh2;
        }
      }
      {  // alignment of the two key-point estimates (weight 0 in full.yaml)
        const float ex = a.xyz_fk[o] - a.xyz_int[o], ey = a.xyz_fk[o + 1] - a.xyz_int[o + 1], ez = a.xyz_fk[o + 2] - a.xyz_int[o + 2];
        const float n = sqrtf(ex * ex + ey * ey + ez * ez);
        t[9] += n;
        if (n > 0.f && a.w[9] != 0.f) {
          const float c = a.w[9] / (n * B * J);
          dfk[0] += c * ex; dfk[1] += c * ey; dfk[2] += c * ez;
          dint[0] -= c * ex; dint[1] -= c * ey; dint[2] -= c * ez;
        }
      }
      if (grad) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { a.d_xyz_fk[o + k] = dfk[k]; a.d_xyz_int[o + k] = dint[k]; }
      }
    }
  }
  const float norm[10] = {1.f / (B * P), 1.f / (B * a.R), 1.f / n_root, 1.f / B, 1.f / B, 1.f / (B * J), 1.f / n_valid, 1.f / n_valid,
                          1.f / (B * J), 1.f / (B * J)};
  // weights in term order: joint, rot, uv, depth, trans, error3d, error2d, error2d_int, error3d_int, align
  const float wt[10] = {a.w[0], a.w[1], a.w[2], a.w[3], a.w[4], a.w[6], a.w[5], a.w[7], a.w[8], a.w[9]};
  float total = 0.f;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const float v = block_sum(t[k], red) * norm[k];
    if (threadIdx.x == 0) a.out[k] = v;
    total += wt[k] * v;
  }
  if (threadIdx.x == 0) a.out[10] = total;
}

}  // namespace hrp

using namespace hrp;

extern "C" int hrp_pose_loss(const hrp_pose_loss_desc* d, void* stream) {
  HRP_REQUIRE(d && d->pose && d->rot && d->trans && d->root_uv && d->depth && d->xyz_int && d->xyz_fk, "pose_loss: null prediction");
  HRP_REQUIRE(d->gt_pose && d->gt_root_rot && d->gt_root_trans && d->gt_root_uv && d->gt_kp3d && d->gt_kp2d && d->mask && d->K && d->out,
              "pose_loss: null ground truth / output");
  HRP_REQUIRE(d->B > 0 && d->P > 0 && d->J > 0 && d->root >= 0 && d->root < d->J && d->image_size > 0.f, "pose_loss: sizes");
  const bool g = d->d_pose != nullptr;
  HRP_REQUIRE(!g || (d->d_rot && d->d_trans && d->d_root_uv && d->d_depth && d->d_xyz_int && d->d_xyz_fk), "pose_loss: all gradient buffers or none");
  LossArgs a;
  a.pose = d->pose; a.rot = d->rot; a.trans = d->trans; a.root_uv = d->root_uv; a.depth = d->depth; a.xyz_int = d->xyz_int; a.xyz_fk = d->xyz_fk;
  a.g_pose = d->gt_pose; a.g_rot = d->gt_root_rot; a.g_trans = d->gt_root_trans; a.g_root_uv = d->gt_root_uv; a.g_kp3d = d->gt_kp3d;
  a.g_kp2d = d->gt_kp2d; a.mask = d->mask; a.K = d->K;
  a.d_pose = d->d_pose; a.d_rot = d->d_rot; a.d_trans = d->d_trans; a.d_root_uv = d->d_root_uv; a.d_depth = d->d_depth;
  a.d_xyz_int = d->d_xyz_int; a.d_xyz_fk = d->d_xyz_fk;
  a.out = d->out;
  for (int k = 0; k < 10; ++k) a.w[k] = d->weights[k];
  a.B = d->B; a.P = d->P; a.J = d->J; a.root = d->root; a.S = d->image_size;
  a.R = d->rot_dim == 0 ? 6 : d->rot_dim;
  HRP_REQUIRE(a.R == 6 || a.R == 4, "pose_loss: rot_dim=%d", d->rot_dim);
  hipLaunchKernelGGL(pose_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("pose_loss");
}

extern "C" int hrp_project_fwd(const float* K, const float* pts, int B, int P, float* uv, void* stream) {
  HRP_REQUIRE(K && pts && uv && B > 0 && P > 0, "project_fwd: bad args");
  hipLaunchKernelGGL(project_fwd_kernel, dim3(cdiv(B * P, 64)), dim3(64), 0, (hipStream_t)stream, K, pts, B, P, uv);
  return check_launch("project_fwd");
}
extern "C" int hrp_project_bwd(const float* K, const float* pts, const float* duv, int B, int P, float* dpts, void* stream) {
  HRP_REQUIRE(K && pts && duv && dpts && B > 0 && P > 0, "project_bwd: bad args");
  hipLaunchKernelGGL(project_bwd_kernel, dim3(cdiv(B * P, 64)), dim3(64), 0, (hipStream_t)stream, K, pts, duv, B, P, dpts);
  return check_launch("project_bwd");
}

// ---- flat soft-argmax (HeatmapIntegralJoint, integral.py:206-232): per (sample, channel) softmax over the H*W positions
// of a small map, coord = E[flat index] / (H*W) in [0, 1).  One wave per (channel, sample); ms keeps (max, sum).
template <typename T>
__global__ __launch_bounds__(64) void softargmax_flat_fwd_kernel(const void* __restrict__ logits, int HW, int pitch,
                                                                 float* __restrict__ coord, float* __restrict__ ms) {
  const int j = blockIdx.x, b = blockIdx.y, J = gridDim.x, lane = threadIdx.x;
  float m = -INFINITY;
  for (int p = lane; p < HW; p += 64) m = fmaxf(m, Elem<T>::ld(logits, ((size_t)b * HW + p) * pitch + j));
  m = wave_max(m);
  float s = 0.f, sx = 0.f;
  for (int p = lane; p < HW; p += 64) {
    const float e = __expf(Elem<T>::ld(logits, ((size_t)b * HW + p) * pitch + j) - m);
    s += e; sx += e * (float)p;
  }
  s = wave_sum(s); sx = wave_sum(sx);
  if (lane == 0) {
    coord[b * J + j] = sx / s / (float)HW;
    ms[(b * J + j) * 2] = m; ms[(b * J + j) * 2 + 1] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(64) void softargmax_flat_bwd_kernel(const void* __restrict__ logits, int HW, int pitch,
                                                                 const float* __restrict__ coord, const float* __restrict__ ms,
                                                                 const float* __restrict__ dcoord, void* __restrict__ dlogits, int dpitch) {
  const int j = blockIdx.x, b = blockIdx.y, J = gridDim.x, lane = threadIdx.x;
  const float m = ms[(b * J + j) * 2], s = ms[(b * J + j) * 2 + 1], c = coord[b * J + j], g = dcoord[b * J + j];
  for (int p = lane; p < HW; p += 64) {
    const float pr = __expf(Elem<T>::ld(logits, ((size_t)b * HW + p) * pitch + j) - m) / s;
    Elem<T>::st(dlogits, ((size_t)b * HW + p) * dpitch + j, pr * ((float)p / (float)HW - c) * g);
  }
}

extern "C" int hrp_softargmax_flat_fwd(const void* logits, int dtype, int B, int J, int HW, int pitch, float* coord, float* ms,
                                       void* stream) {
  HRP_REQUIRE(logits && coord && ms && B > 0 && J > 0 && HW > 0 && pitch >= J, "softargmax_flat_fwd: bad args");
  if (dtype == HRP_F32) hipLaunchKernelGGL(softargmax_flat_fwd_kernel<float>, dim3(J, B), dim3(64), 0, (hipStream_t)stream, logits, HW, pitch, coord, ms);
  else hipLaunchKernelGGL(softargmax_flat_fwd_kernel<bf16_t>, dim3(J, B), dim3(64), 0, (hipStream_t)stream, logits, HW, pitch, coord, ms);
  return check_launch("softargmax_flat_fwd");
}

extern "C" int hrp_softargmax_flat_bwd(const void* logits, int dtype, int B, int J, int HW, int pitch, const float* coord,
                                       const float* ms, const float* dcoord, void* dlogits, int dpitch, void* stream) {
  HRP_REQUIRE(logits && coord && ms && dcoord && dlogits && B > 0 && J > 0 && HW > 0, "softargmax_flat_bwd: bad args");
  if (dtype == HRP_F32) hipLaunchKernelGGL(softargmax_flat_bwd_kernel<float>, dim3(J, B), dim3(64), 0, (hipStream_t)stream, logits, HW, pitch, coord, ms, dcoord, dlogits, dpitch);
  else hipLaunchKernelGGL(softargmax_flat_bwd_kernel<bf16_t>, dim3(J, B), dim3(64), 0, (hipStream_t)stream, logits, HW, pitch, coord, ms, dcoord, dlogits, dpitch);
  return check_launch("softargmax_flat_bwd");
}

extern "C" int hrp_softargmax3d_fwd(const void* logits, int dtype, int B, int J, int D, int H, int W, int pitch,
                                    int root, int fix_root, float* uvd, float* ms, void* stream) {
  HRP_REQUIRE(logits && uvd && ms && B > 0 && J > 0, "softargmax_fwd: bad args");
  const int vec = dtype == HRP_F32 ? 4 : 8;
  HRP_REQUIRE(D % vec == 0 && D / vec <= 256 && 256 % (D / vec) == 0, "softargmax: D=%d unsupported", D);
  HRP_REQUIRE(pitch % vec == 0 && (uintptr_t)logits % 16 == 0, "softargmax: alignment");
  dim3 grid(J, B);
  if (dtype == HRP_F32) hipLaunchKernelGGL(softargmax_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, logits, J, D, H, W, pitch, root, fix_root, uvd, ms);
  else hipLaunchKernelGGL(softargmax_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, logits, J, D, H, W, pitch, root, fix_root, uvd, ms);
  return check_launch("softargmax_fwd");
}

extern "C" int hrp_softargmax3d_bwd(const void* logits, int dtype, int B, int J, int D, int H, int W, int pitch,
                                    int root, int fix_root, const float* uvd, const float* ms, const float* duvd,
                                    void* dlogits, int dpitch, void* stream) {
  HRP_REQUIRE(logits && uvd && ms && duvd && dlogits, "softargmax_bwd: bad args");
  const int vec = dtype == HRP_F32 ? 4 : 8;
  HRP_REQUIRE(D % vec == 0 && pitch % vec == 0 && dpitch % vec == 0, "softargmax_bwd: alignment");
  long total = (long)H * W * (J * D / vec);
  int gx = (int)((total + 255) / 256);
  if (gx > 64) gx = 64;
  dim3 grid(gx, B);
  if (dtype == HRP_F32) hipLaunchKernelGGL(softargmax_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, logits, J, D, H, W, pitch, root, fix_root, uvd, ms, duvd, dlogits, dpitch);
  else hipLaunchKernelGGL(softargmax_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, logits, J, D, H, W, pitch, root, fix_root, uvd, ms, duvd, dlogits, dpitch);
  return check_launch("softargmax_bwd");
}

extern "C" int hrp_pose_geometry_fwd(const float* gamma, const float* k_value, const float* uvd, const float* K,
                                     int B, int J, int root, float image_size, float depth_factor,
                                     float* depth, float* xyz, float* root_uv, float* trans, void* stream) {
  HRP_REQUIRE(gamma && k_value && uvd && K && depth && xyz && root_uv && trans && B > 0, "pose_geometry_fwd: bad args");
  hipLaunchKernelGGL(pose_geometry_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, gamma, k_value, uvd, K, B, J, root,
                     image_size, depth_factor, depth, xyz, root_uv, trans);
  return check_launch("pose_geometry_fwd");
}

extern "C" int hrp_pose_geometry_bwd(const float* gamma, const float* k_value, const float* uvd, const float* K,
                                     int B, int J, int root, float image_size, float depth_factor,
                                     const float* d_depth, const float* d_xyz, const float* d_root_uv, const float* d_trans,
                                     float* d_gamma, float* d_uvd, void* stream) {
  HRP_REQUIRE(gamma && k_value && uvd && K && d_gamma && d_uvd && B > 0, "pose_geometry_bwd: bad args");
  hipLaunchKernelGGL(pose_geometry_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, gamma, k_value, uvd, K, B, J, root, 0,
                     image_size, depth_factor, d_depth, d_xyz, d_root_uv, d_trans, d_gamma, d_uvd);
  return check_launch("pose_geometry_bwd");
}

// ---- rot6d composition (rot_iterative_matmul, full_net.py:346-362): out = rot6d(R(a) R(b)) ---------------------------
// forward: one thread per sample; backward: the same templated code on dual numbers, one evaluation per input (12).
template <class S>
__device__ __forceinline__ void rot6d_compose(const S* a, const S* b, S* out) {
  const S z3[3] = {lift<S>(0.f), lift<S>(0.f), lift<S>(0.f)};
  const Rigid<S> A = base_to_cam<S>(a, z3), B = base_to_cam<S>(b, z3);
#pragma unroll
  for (int i = 0; i < 2; ++i)      // geometries.py:117-131: the first two rows of the product
#pragma unroll
    for (int j = 0; j < 3; ++j) out[3 * i + j] = A.r[i][0] * B.r[0][j] + A.r[i][1] * B.r[1][j] + A.r[i][2] * B.r[2][j];
}

__global__ void rot6d_compose_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float av[6], bv[6], o[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { av[i] = a[n * 6 + i]; bv[i] = b[n * 6 + i]; }
  rot6d_compose<float>(av, bv, o);
#pragma unroll
  for (int i = 0; i < 6; ++i) out[n * 6 + i] = o[i];
}

__global__ void rot6d_compose_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ dout,
                                         float* __restrict__ da, float* __restrict__ db, int N, int acc_a, int acc_b) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  Dual av[6], bv[6], o[6];
  float g[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { av[i] = {a[n * 6 + i], 0.f}; bv[i] = {b[n * 6 + i], 0.f}; g[i] = dout[n * 6 + i]; }
  for (int k = 0; k < 12; ++k) {
    if (k < 6) av[k].d = 1.f; else bv[k - 6].d = 1.f;
    rot6d_compose<Dual>(av, bv, o);
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) t += o[i].d * g[i];
    if (k < 6) { av[k].d = 0.f; if (da) da[n * 6 + k] = acc_a ? da[n * 6 + k] + t : t; }
    else { bv[k - 6].d = 0.f; if (db) db[n * 6 + k - 6] = acc_b ? db[n * 6 + k - 6] + t : t; }
  }
}

extern "C" int hrp_rot6d_compose_fwd(const float* a, const float* b, float* out, int N, void* stream) {
  HRP_REQUIRE(a && b && out && N > 0, "rot6d_compose_fwd: bad args");
  hipLaunchKernelGGL(rot6d_compose_fwd_kernel, dim3(cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, a, b, out, N);
  return check_launch("rot6d_compose_fwd");
}

extern "C" int hrp_rot6d_compose_bwd(const float* a, const float* b, const float* dout, float* da, float* db, int N, int acc_a,
                                     int acc_b, void* stream) {
  HRP_REQUIRE(a && b && dout && (da || db) && N > 0, "rot6d_compose_bwd: bad args");
  hipLaunchKernelGGL(rot6d_compose_bwd_kernel, dim3(cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, a, b, dout, da, db, N, acc_a, acc_b);
  return check_launch("rot6d_compose_bwd");
}

namespace hrp {
// ---- mesh posing (mesh_renderer.py:126-173 + urdf_robot.py:242-275): one workgroup row per sample ------------------------
__global__ __launch_bounds__(256) void mesh_pose_kernel(const hrp_fk_chain* __restrict__ ch, const float* __restrict__ q,
                                                        const float* __restrict__ r6, const float* __restrict__ tr, int root_kp,
                                                        const float* __restrict__ verts, const uint8_t* __restrict__ vert_link, int V,
                                                        const float* __restrict__ K, float* __restrict__ xyz, float* __restrict__ uv) {
  __shared__ float P[HRP_FK_MAX_KP][12];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int nl = ch->nkp;
  if (tid < nl) {
    const int dof = ch->dof;
    float ql[HRP_FK_MAX_JOINTS];
    for (int i = 0; i < dof; ++i) ql[i] = q[(size_t)b * dof + i];
    float rl[6], tl[3];
    for (int i = 0; i < 6; ++i) rl[i] = r6[6 * b + i];
    for (int i = 0; i < 3; ++i) tl[i] = tr[3 * b + i];
    Rigid<float> M = base_to_cam<float>(rl, tl);
    if (root_kp >= 0) M = rigid_mul(M, rigid_inverse(frame_pose<float>(ch, ch->kp_frame[root_kp], ql)));
    if (M.t[2] < 0.f) {      // urdf_robot.py:250-253: the camera looks along +z; a pose behind it is mirrored through the origin
      for (int i = 0; i < 3; ++i) { M.t[i] = -M.t[i]; for (int k = 0; k < 3; ++k) M.r[i][k] = -M.r[i][k]; }
    }
    const Rigid<float> T = rigid_mul(M, frame_pose<float>(ch, ch->kp_frame[tid], ql));
    for (int i = 0; i < 3; ++i) { P[tid][4 * i] = T.r[i][0]; P[tid][4 * i + 1] = T.r[i][1]; P[tid][4 * i + 2] = T.r[i][2]; P[tid][4 * i + 3] = T.t[i]; }
  }
  __syncthreads();
  for (int v = blockIdx.x * 256 + tid; v < V; v += gridDim.x * 256) {
    const float* T = P[vert_link[v]];
    const float x = verts[3 * v], y = verts[3 * v + 1], z = verts[3 * v + 2];
    float p[3];
    for (int i = 0; i < 3; ++i) p[i] = T[4 * i] * x + T[4 * i + 1] * y + T[4 * i + 2] * z + T[4 * i + 3];
    float* o = xyz + ((size_t)b * V + v) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    if (K && uv) {
      float w[2];
      project<float>(K + 9 * b, p, w);
      uv[((size_t)b * V + v) * 2] = w[0];
      uv[((size_t)b * V + v) * 2 + 1] = w[1];
    }
  }
}
}  // namespace hrp

namespace hrp {
// d (rot6d, trans) of sum(d_xyz * xyz) for hrp_mesh_pose's output: xyz[v] = s (R(rot6d) w_v + trans), w_v = T_root^-1 T_link(v) verts[v]
// (independent of rot6d / trans; the joint angles are detached on this path, urdf_robot.py:267), s = -1 for a mirrored sample.
// One workgroup per sample: G_t = sum g_v, G_R = sum g_v w_v^T reduced in a fixed order, then nine lanes differentiate the
// rotation matrix in forward mode.
__global__ __launch_bounds__(256) void mesh_pose_bwd_kernel(const hrp_fk_chain* __restrict__ ch, const float* __restrict__ q,
                                                            const float* __restrict__ r6, const float* __restrict__ tr, int root_kp,
                                                            const float* __restrict__ verts, const uint8_t* __restrict__ vert_link, int V,
                                                            const float* __restrict__ d_xyz, float* __restrict__ d_r6, float* __restrict__ d_tr) {
  __shared__ float P[HRP_FK_MAX_KP][12];
  __shared__ float red[256][12];
  __shared__ float flip;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nl = ch->nkp, dof = ch->dof;
  float rl[6], tl[3];
  for (int i = 0; i < 6; ++i) rl[i] = r6[6 * b + i];
  for (int i = 0; i < 3; ++i) tl[i] = tr[3 * b + i];
  if (tid < nl || tid == 255) {
    float ql[HRP_FK_MAX_JOINTS];
    for (int i = 0; i < dof; ++i) ql[i] = q[(size_t)b * dof + i];
    Rigid<float> Rinv = rigid_identity<float>();
    if (root_kp >= 0) Rinv = rigid_inverse(frame_pose<float>(ch, ch->kp_frame[root_kp], ql));
    if (tid == 255) {
      const Rigid<float> M = rigid_mul(base_to_cam<float>(rl, tl), Rinv);
      flip = M.t[2] < 0.f ? -1.f : 1.f;
    } else {
      const Rigid<float> T = rigid_mul(Rinv, frame_pose<float>(ch, ch->kp_frame[tid], ql));
      for (int i = 0; i < 3; ++i) { P[tid][4 * i] = T.r[i][0]; P[tid][4 * i + 1] = T.r[i][1]; P[tid][4 * i + 2] = T.r[i][2]; P[tid][4 * i + 3] = T.t[i]; }
    }
  }
  __syncthreads();
  float acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int v = tid; v < V; v += 256) {
    const float* T = P[vert_link[v]];
    const float x = verts[3 * v], y = verts[3 * v + 1], z = verts[3 * v + 2];
    float w[3];
    for (int i = 0; i < 3; ++i) w[i] = T[4 * i] * x + T[4 * i + 1] * y + T[4 * i + 2] * z + T[4 * i + 3];
    const float* g = d_xyz + ((size_t)b * V + v) * 3;
    for (int i = 0; i < 3; ++i) {
      acc[9 + i] += g[i];
      for (int j = 0; j < 3; ++j) acc[3 * i + j] += g[i] * w[j];
    }
  }
  for (int i = 0; i < 12; ++i) red[tid][i] = acc[i];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st)
      for (int i = 0; i < 12; ++i) red[tid][i] += red[tid + st][i];
    __syncthreads();
  }
  if (tid < 9) {
    Dual rd[6], td[3];
    for (int i = 0; i < 6; ++i) rd[i] = {rl[i], tid == i ? 1.f : 0.f};
    for (int i = 0; i < 3; ++i) td[i] = {tl[i], tid == 6 + i ? 1.f : 0.f};
    const Rigid<Dual> M = base_to_cam<Dual>(rd, td);
    float g = 0.f;
    for (int i = 0; i < 3; ++i) {
      g += red[0][9 + i] * M.t[i].d;
      for (int j = 0; j < 3; ++j) g += red[0][3 * i + j] * M.r[i][j].d;
    }
    g *= flip;
    if (tid < 6) d_r6[6 * b + tid] = g;
    else d_tr[3 * b + tid - 6] = g;
  }
}
}  // namespace hrp

extern "C" int hrp_mesh_pose_bwd(const hrp_fk_chain* chain_dev, const float* q, const float* rot6d, const float* trans, int B, int root_kp,
                                 const float* verts, const uint8_t* vert_link, int V, const float* d_xyz, float* d_rot6d, float* d_trans,
                                 void* stream) {
  using namespace hrp;
  HRP_REQUIRE(chain_dev && q && rot6d && trans && verts && vert_link && d_xyz && d_rot6d && d_trans && B > 0 && V > 0, "mesh_pose_bwd: bad args");
  HRP_REQUIRE(root_kp < HRP_FK_MAX_KP, "mesh_pose_bwd: root_kp=%d", root_kp);
  hipLaunchKernelGGL(mesh_pose_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, chain_dev, q, rot6d, trans, root_kp, verts, vert_link, V,
                     d_xyz, d_rot6d, d_trans);
  return check_launch("mesh_pose_bwd");
}

extern "C" int hrp_mesh_pose(const hrp_fk_chain* chain_dev, const float* q, const float* rot6d, const float* trans, int B, int root_kp,
                             const float* verts, const uint8_t* vert_link, int V, const float* K, float* xyz, float* uv, void* stream) {
  using namespace hrp;
  HRP_REQUIRE(chain_dev && q && rot6d && trans && verts && vert_link && xyz && B > 0 && V > 0, "mesh_pose: bad args");
  HRP_REQUIRE(root_kp < HRP_FK_MAX_KP, "mesh_pose: root_kp=%d", root_kp);
  int gx = (V + 256 * 4 - 1) / (256 * 4);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(mesh_pose_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, chain_dev, q, rot6d, trans, root_kp, verts, vert_link, V, K, xyz, uv);
  return check_launch("mesh_pose");
}

extern "C" int hrp_fk_project_rot_fwd(const hrp_fk_chain* chain_dev, const float* q, const float* rot, int rot_dim, const float* trans,
                                      const float* K, int B, int root, float* xyz, float* uv, float* root_rot, void* stream) {
  HRP_REQUIRE(chain_dev && q && rot && trans && xyz && B > 0, "fk_project_fwd: bad args");
  HRP_REQUIRE(rot_dim == 6 || rot_dim == 4, "fk_project_fwd: rot_dim=%d (6: two rows of the rotation matrix, 4: quaternion w x y z)", rot_dim);
  hipLaunchKernelGGL(fk_project_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, chain_dev, q, rot, trans, K, root, xyz, uv, root_rot, rot_dim);
  return check_launch("fk_project_fwd");
}
extern "C" int hrp_fk_project_fwd(const hrp_fk_chain* chain_dev, const float* q, const float* rot6d, const float* trans,
                                  const float* K, int B, int root, float* xyz, float* uv, float* root_rot6d, void* stream) {
  return hrp_fk_project_rot_fwd(chain_dev, q, rot6d, 6, trans, K, B, root, xyz, uv, root_rot6d, stream);
}

extern "C" int hrp_fk_project_rot_bwd(const hrp_fk_chain* chain_dev, const float* q, const float* rot, int rot_dim, const float* trans,
                                      const float* K, int B, int root, const float* d_xyz, const float* d_uv,
                                      float* d_q, float* d_rot, float* d_trans, void* stream) {
  HRP_REQUIRE(chain_dev && q && rot && trans && d_q && d_rot && d_trans && B > 0, "fk_project_bwd: bad args");
  HRP_REQUIRE(rot_dim == 6 || rot_dim == 4, "fk_project_bwd: rot_dim=%d", rot_dim);
  hipLaunchKernelGGL(fk_project_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, chain_dev, q, rot, trans, K, root, d_xyz, d_uv, d_q, d_rot, d_trans, rot_dim);
  return check_launch("fk_project_bwd");
}
extern "C" int hrp_fk_project_bwd(const hrp_fk_chain* chain_dev, const float* q, const float* rot6d, const float* trans,
                                  const float* K, int B, int root, const float* d_xyz, const float* d_uv,
                                  float* d_q, float* d_rot6d, float* d_trans, void* stream) {
  return hrp_fk_project_rot_bwd(chain_dev, q, rot6d, 6, trans, K, B, root, d_xyz, d_uv, d_q, d_rot6d, d_trans, stream);
}

// ---- L1 loss of the DepthNet trainer (reference scripts/train_depthnet.py:231-250: L1Loss(model(images, k) / 1000, gt_depth)):
// loss = mean |pred * scale - gt|, d_pred = sign(pred * scale - gt) * scale / n.  One workgroup (n = batch x 1 values).
namespace hrp {
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt, float scale, int n,
                                                       float* __restrict__ loss, float* __restrict__ d_pred) {
  __shared__ float part[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float e = pred[i] * scale - gt[i];
    s += fabsf(e);
    if (d_pred) d_pred[i] = (e > 0.f ? 1.f : e < 0.f ? -1.f : 0.f) * scale / (float)n;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (part[0] + part[1] + part[2] + part[3]) / (float)n;
}
}  // namespace hrp

namespace hrp {
// ---- mask losses of the self-supervised trainer (scripts/train_sim2real.py:435-468) ------------------------------------
// pass 1: per image I = sum s r, S = sum s, R = sum r, M = the mask term's sum; one workgroup per image, fixed order
__global__ __launch_bounds__(256) void s2r_sums_kernel(const hrp_sim2real_loss_desc d) {
  const int b = blockIdx.x;
  const float* r = d.rendered + (size_t)b * d.HW;
  const float* s = d.seg + (size_t)b * d.HW;
  float aI = 0.f, aS = 0.f, aR = 0.f, aM = 0.f;
  for (int p = threadIdx.x; p < d.HW; p += 256) {
    const float rv = r[p], sv = s[p];
    aI += sv * rv; aS += sv; aR += rv;
    if (d.mask_loss == 1) aM -= sv * fmaxf(logf(rv), -100.f) + (1.f - sv) * fmaxf(logf(1.f - rv), -100.f);
    else { const float e = rv - sv; aM += e * e; }
  }
  __shared__ float part[4][4];
  aI = wave_sum(aI); aS = wave_sum(aS); aR = wave_sum(aR); aM = wave_sum(aM);
  if ((threadIdx.x & 63) == 0) { float* q = part[threadIdx.x >> 6]; q[0] = aI; q[1] = aS; q[2] = aR; q[3] = aM; }
  __syncthreads();
  if (threadIdx.x < 4) d.workspace[b * 8 + threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// pass 2 (one workgroup): the terms, the per-image gradient coefficients, the key-point gradients
__global__ __launch_bounds__(256) void s2r_finalize_kernel(const hrp_sim2real_loss_desc d) {
  __shared__ float sh_align;
  const int n3 = d.B * d.K;
  // align: every thread its own points, the sum serially by thread 0 (B * K is a few hundred)
  if (threadIdx.x == 0) {
    float mask = 0.f, iou = 0.f, num = 0.f, nf = 0.f;
    for (int b = 0; b < d.B; ++b) {
      const float* w = d.workspace + b * 8;
      const float I = w[0], S = w[1], R = w[2];
      mask += w[3];
      const float U = S + R - I;
      iou += I / U;
      const float so = S - I, ro = R - I, ratio = so / ro;
      const float f = (ratio > 5.0f || ratio < 0.2f) ? 1.f : 0.f;
      num += fabsf(logf(ratio)) * f;
      nf += f;
    }
    const float npx = (float)d.B * (float)d.HW;
    const float l_mask = d.mask_loss == 2 ? 0.001f * mask : mask / npx;
    const float l_iou = 1.f - iou / (float)d.B;
    const float l_scale = num / (nf + 1e-9f);
    float al = 0.f;
    for (int i = 0; i < n3; ++i) {
      const float ex = d.kp3d[3 * i] - d.kp3d_int[3 * i], ey = d.kp3d[3 * i + 1] - d.kp3d_int[3 * i + 1], ez = d.kp3d[3 * i + 2] - d.kp3d_int[3 * i + 2];
      al += sqrtf(ex * ex + ey * ey + ez * ez);
    }
    const float l_align = al / (float)n3;
    d.terms[1] = l_mask; d.terms[2] = l_iou; d.terms[3] = l_scale; d.terms[4] = l_align;
    d.terms[0] = d.w_mask * l_mask + d.w_iou * l_iou + d.w_scale * l_scale + d.w_align * l_align;
    sh_align = nf;
  }
  __syncthreads();
  const float nf = sh_align;
  // d loss / d rendered[b, p] = A_b * s + B_b * (1 - s) + the mask term's own (pass 3)
  for (int b = threadIdx.x; b < d.B; b += 256) {
    float* w = d.workspace + b * 8;
    const float I = w[0], S = w[1], R = w[2];
    const float U = S + R - I, so = S - I, ro = R - I, ratio = so / ro;
    float A = 0.f, Bc = 0.f;
    if (d.w_iou != 0.f) { A += d.w_iou * (-1.f / ((float)d.B * U)); Bc += d.w_iou * (I / ((float)d.B * U * U)); }
    if (d.w_scale != 0.f && (ratio > 5.0f || ratio < 0.2f)) {
      const float c = d.w_scale * (logf(ratio) >= 0.f ? 1.f : -1.f) / (nf + 1e-9f);
      A += c * (-1.f / so); Bc += c * (-1.f / ro);
    }
    w[4] = A; w[5] = Bc;
  }
  if (d.d_kp3d || d.d_kp3d_int) {
    for (int i = threadIdx.x; i < n3; i += 256) {
      const float ex = d.kp3d[3 * i] - d.kp3d_int[3 * i], ey = d.kp3d[3 * i + 1] - d.kp3d_int[3 * i + 1], ez = d.kp3d[3 * i + 2] - d.kp3d_int[3 * i + 2];
      const float nrm = sqrtf(ex * ex + ey * ey + ez * ez);
      const float k = nrm > 0.f ? d.w_align / (nrm * (float)n3) : 0.f;
      if (d.d_kp3d) { d.d_kp3d[3 * i] = k * ex; d.d_kp3d[3 * i + 1] = k * ey; d.d_kp3d[3 * i + 2] = k * ez; }
      if (d.d_kp3d_int) { d.d_kp3d_int[3 * i] = -k * ex; d.d_kp3d_int[3 * i + 1] = -k * ey; d.d_kp3d_int[3 * i + 2] = -k * ez; }
    }
  }
}

// pass 3: the gradient with respect to the rendered masks
__global__ __launch_bounds__(256) void s2r_grad_kernel(const hrp_sim2real_loss_desc d) {
  const int b = blockIdx.y;
  const float A = d.workspace[b * 8 + 4], Bc = d.workspace[b * 8 + 5];
  const float npx = (float)d.B * (float)d.HW;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < d.HW; p += gridDim.x * 256) {
    const size_t o = (size_t)b * d.HW + p;
    const float rv = d.rendered[o], sv = d.seg[o];
    float g = A * sv + Bc * (1.f - sv);
    if (d.w_mask != 0.f) {
      if (d.mask_loss == 0) g += d.w_mask * 2.f * (rv - sv) / npx;
      else if (d.mask_loss == 2) g += d.w_mask * 0.002f * (rv - sv);
      else g += d.w_mask * (rv - sv) / fmaxf((1.f - rv) * rv, 1e-12f) / npx;
    }
    d.d_rendered[o] = g;
  }
}
}  // namespace hrp

extern "C" int hrp_sim2real_loss(const hrp_sim2real_loss_desc* dp, void* stream) {
  using namespace hrp;
  HRP_REQUIRE(dp && dp->rendered && dp->seg && dp->kp3d && dp->kp3d_int && dp->terms && dp->workspace, "sim2real_loss: null pointer");
  HRP_REQUIRE(dp->B > 0 && dp->HW > 0 && dp->K > 0 && dp->mask_loss >= 0 && dp->mask_loss <= 2, "sim2real_loss: B=%d HW=%d K=%d mask_loss=%d",
              dp->B, dp->HW, dp->K, dp->mask_loss);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(s2r_sums_kernel, dim3(dp->B), dim3(256), 0, s, *dp);
  hipLaunchKernelGGL(s2r_finalize_kernel, dim3(1), dim3(256), 0, s, *dp);
  if (dp->d_rendered) {
    const int gx = (dp->HW + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(s2r_grad_kernel, dim3(gx > 0 ? gx : 1, dp->B), dim3(256), 0, s, *dp);
  }
  return check_launch("sim2real_loss");
}

extern "C" int hrp_l1_loss(const float* pred, const float* gt, float scale, int n, float* loss, float* d_pred, void* stream) {
  using namespace hrp;
  HRP_REQUIRE(pred && gt && loss && n > 0, "l1_loss: bad arguments");
  hipLaunchKernelGGL(l1_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, gt, scale, n, loss, d_pred);
  return check_launch("l1_loss_kernel");
}
