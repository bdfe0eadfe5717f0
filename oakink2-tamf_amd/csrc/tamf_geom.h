// Geometry kernels either side of the trunks: pose decode (rot6d -> rotation matrix ->
// unit quaternion) and the hand-vertex -> nearest object point distance feature of the refiner.  fp32, VALU-bound.
#pragma once
#include "tamf_device.h"

TAMF_DEV void gs_rows(const float* a, float (&r)[9]) {
  // Gram-Schmidt of (a[0..2], a[3..5]); rows of the rotation matrix = b1, b2, b1 x b2
  // (dev_fn/transform/rotation.py:446-467; F.normalize: v / max(||v||, 1e-12))
  const float n1 = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), 1e-12f);
  const float b1x = a[0] / n1, b1y = a[1] / n1, b1z = a[2] / n1;
  const float dp = b1x * a[3] + b1y * a[4] + b1z * a[5];
  float b2x = a[3] - dp * b1x, b2y = a[4] - dp * b1y, b2z = a[5] - dp * b1z;
  const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
  b2x /= n2; b2y /= n2; b2z /= n2;
  r[0] = b1x; r[1] = b1y; r[2] = b1z;
  r[3] = b2x; r[4] = b2y; r[5] = b2z;
  r[6] = b1y * b2z - b1z * b2y;
  r[7] = b1z * b2x - b1x * b2z;
  r[8] = b1x * b2y - b1y * b2x;
}

// pose_repr (N, F) with F = 3 + 6*J  ->  tsl (N, 3), quat (N, J, 4) in (w, x, y, z) with w >= 0
// (oakink2_tamf/launch/sample_refine.py:254-260; rotation.py:167-213 rotmat_to_quat, :24-35 standardize_quat)
__global__ void pose_decode_kernel(const float* __restrict__ pose, float* __restrict__ tsl, float* __restrict__ quat, long N,
                                   int J, int F) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * J) return;
  const int j = (int)(idx % J);
  const long n = idx / J;
  const float* p = pose + n * F;
  if (j == 0 && tsl) {
    tsl[n * 3 + 0] = p[0];
    tsl[n * 3 + 1] = p[1];
    tsl[n * 3 + 2] = p[2];
  }
  float a[6], m[9];
#pragma unroll
  for (int k = 0; k < 6; ++k) a[k] = p[3 + j * 6 + k];
  gs_rows(a, m);
  const float s0 = 1.0f + m[0] + m[4] + m[8], s1 = 1.0f + m[0] - m[4] - m[8];
  const float s2 = 1.0f - m[0] + m[4] - m[8], s3 = 1.0f - m[0] - m[4] + m[8];
  const float q0 = s0 > 0.f ? sqrtf(s0) : 0.f, q1 = s1 > 0.f ? sqrtf(s1) : 0.f;
  const float q2 = s2 > 0.f ? sqrtf(s2) : 0.f, q3 = s3 > 0.f ? sqrtf(s3) : 0.f;
  // argmax (first maximum, as torch.argmax) picks the best-conditioned candidate
  int best = 0;
  float qb = q0;
  if (q1 > qb) { qb = q1; best = 1; }
  if (q2 > qb) { qb = q2; best = 2; }
  if (q3 > qb) { qb = q3; best = 3; }
  float c[4];
  if (best == 0) { c[0] = q0 * q0; c[1] = m[7] - m[5]; c[2] = m[2] - m[6]; c[3] = m[3] - m[1]; }
  else if (best == 1) { c[0] = m[7] - m[5]; c[1] = q1 * q1; c[2] = m[3] + m[1]; c[3] = m[2] + m[6]; }
  else if (best == 2) { c[0] = m[2] - m[6]; c[1] = m[3] + m[1]; c[2] = q2 * q2; c[3] = m[5] + m[7]; }
  else { c[0] = m[3] - m[1]; c[1] = m[6] + m[2]; c[2] = m[7] + m[5]; c[3] = q3 * q3; }
  const float den = 2.0f * fmaxf(qb, 0.1f);
  float w = c[0] / den, x = c[1] / den, y = c[2] / den, z = c[3] / den;
  if (w < 0.f) { w = -w; x = -x; y = -y; z = -z; }
  float* o = quat + idx * 4;
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

// h2o[b, t, v] = min over real objects o and points j of || hand[b,t,v] - (R(b,o,t) pts[b,o,j] + tsl(b,o,t)) ||
// (oakink2_tamf/model/segment_refine_model.py:142-168, model/loss/chamfer_distance.py:4-64 with y_normals = None).
// One workgroup per (t, b): every thread keeps up to 4 hand vertices in registers; object points are transformed to
// the frame's pose on the fly and streamed through LDS in tiles of 256, read back as 16-byte broadcasts.
// frame_min[b, t] (optional) = min over v of h2o[b, t, v]: the per-frame contact distance of the Contact-Ratio score
// (script/compute_score/compute_score_cr.py:122-149: transf_merge_obj_pointcloud + torch.cdist(...).min per frame).
constexpr int H2O_VPT = 4;
__global__ __launch_bounds__(256) void h2o_dist_kernel(const float* __restrict__ hand, const float* __restrict__ traj,
                                                       const float* __restrict__ pts, const int* __restrict__ obj_num,
                                                       float* __restrict__ out, float* __restrict__ frame_min, int T,
                                                       int V, int nobj, int P) {
  __shared__ float4 tile[256];
  __shared__ float wmin[4];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* hv = hand + ((long)b * T + t) * V * 3;
  float vx[H2O_VPT], vy[H2O_VPT], vz[H2O_VPT], best[H2O_VPT];
#pragma unroll
  for (int i = 0; i < H2O_VPT; ++i) {
    const int v = tid + 256 * i;
    const bool ok = v < V;
    vx[i] = ok ? hv[v * 3 + 0] : 0.f;
    vy[i] = ok ? hv[v * 3 + 1] : 0.f;
    vz[i] = ok ? hv[v * 3 + 2] : 0.f;
    best[i] = 3.0e38f;
  }
  const int n = obj_num ? min(obj_num[b], nobj) : nobj;
  for (int o = 0; o < n; ++o) {
    const float* tr = traj + (((long)b * nobj + o) * T + t) * 9;
    float a[6], R[9];
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = tr[3 + k];
    gs_rows(a, R);
    const float tx = tr[0], ty = tr[1], tz = tr[2];
    const float* pp = pts + ((long)b * nobj + o) * P * 3;
    for (int p0 = 0; p0 < P; p0 += 256) {
      const int j = p0 + tid;
      float4 q = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);  // padding point: far away, never the minimum
      if (j < P) {
        const float px = pp[j * 3 + 0], py = pp[j * 3 + 1], pz = pp[j * 3 + 2];
        q.x = fmaf(R[2], pz, fmaf(R[1], py, R[0] * px)) + tx;
        q.y = fmaf(R[5], pz, fmaf(R[4], py, R[3] * px)) + ty;
        q.z = fmaf(R[8], pz, fmaf(R[7], py, R[6] * px)) + tz;
      }
      __syncthreads();
      tile[tid] = q;
      __syncthreads();
#pragma unroll 8
      for (int k = 0; k < 256; ++k) {
        const float4 c = tile[k];
#pragma unroll
        for (int i = 0; i < H2O_VPT; ++i) {
          const float dx = vx[i] - c.x, dy = vy[i] - c.y, dz = vz[i] - c.z;
          best[i] = fminf(best[i], fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        }
      }
    }
  }
  float fm = 3.0e38f;
#pragma unroll
  for (int i = 0; i < H2O_VPT; ++i) {
    const int v = tid + 256 * i;
    if (v < V) {
      const float dv = sqrtf(best[i]);
      if (out) out[((long)b * T + t) * V + v] = dv;
      fm = fminf(fm, dv);
    }
  }
  if (frame_min) {
    fm = wave_reduce<RedMin>(fm);
    if ((tid & 63) == 0) wmin[tid >> 6] = fm;
    __syncthreads();
    if (tid == 0) frame_min[(long)b * T + t] = fminf(fminf(wmin[0], wmin[1]), fminf(wmin[2], wmin[3]));
  }
}

// ---- point-in-closed-mesh test of the SIV score ---------------------------------------------------
// dev_fn/external/libmesh/inside_mesh.py:8-149 (+ triangle_hash.pyx, an acceleration structure only), float64 with the
// reference's operation order and no fused multiply-adds, so that the booleans are bit-identical to numpy's.
// Per triangle (rescaled to the 512-grid frame) 16 doubles are prepared once: t3.xy, the 2D edge matrix a00 a01 a10 a11,
// sign/abs of its determinant, the normal's x, y, sign/abs of its z, t1.xy and t1.z * |n_z|.
constexpr int MESH_TC = 16;
__global__ void mesh_prepare_kernel(const double* __restrict__ verts, const int* __restrict__ faces, int F, double sx, double sy,
                                    double sz, double tx, double ty, double tz, double* __restrict__ tc) {
#pragma clang fp contract(off)
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  double t[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double* v = verts + (long)faces[f * 3 + k] * 3;
    t[k][0] = sx * v[0] + tx;
    t[k][1] = sy * v[1] + ty;
    t[k][2] = sz * v[2] + tz;
  }
  const double a00 = t[0][0] - t[2][0], a01 = t[1][0] - t[2][0], a10 = t[0][1] - t[2][1], a11 = t[1][1] - t[2][1];
  const double det = a00 * a11 - a01 * a10;
  const double v1x = t[2][0] - t[0][0], v1y = t[2][1] - t[0][1], v1z = t[2][2] - t[0][2];
  const double v2x = t[1][0] - t[0][0], v2y = t[1][1] - t[0][1], v2z = t[1][2] - t[0][2];
  const double nx = v1y * v2z - v1z * v2y, ny = v1z * v2x - v1x * v2z, nz = v1x * v2y - v1y * v2x;
  const double an = fabs(nz), sn = nz > 0.0 ? 1.0 : (nz < 0.0 ? -1.0 : 0.0);
  double* o = tc + (long)f * MESH_TC;
  o[0] = t[2][0]; o[1] = t[2][1];
  o[2] = a00; o[3] = a01; o[4] = a10; o[5] = a11;
  o[6] = det > 0.0 ? 1.0 : (det < 0.0 ? -1.0 : 0.0);
  o[7] = fabs(det);
  o[8] = nx; o[9] = ny; o[10] = sn; o[11] = an;
  o[12] = t[0][0]; o[13] = t[0][1];
  o[14] = t[0][2] * an;
  o[15] = 0.0;
}

__global__ __launch_bounds__(256) void mesh_contains_kernel(const double* __restrict__ tc, int F, const double* __restrict__ pts,
                                                            long N, double sx, double sy, double sz, double tx, double ty,
                                                            double tz, double res, unsigned char* __restrict__ out) {
#pragma clang fp contract(off)
  constexpr int TILE = 64;
  __shared__ double tile[TILE * MESH_TC];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double qx = 0.0, qy = 0.0, qz = 0.0;
  bool live = false;
  if (i < N) {
    qx = sx * pts[i * 3 + 0] + tx;
    qy = sy * pts[i * 3 + 1] + ty;
    qz = sz * pts[i * 3 + 2] + tz;
    // inside the rescaled bounding box [0, res]^3 (:43) and inside the hash grid (cell index < res, triangle_hash.pyx:63-67)
    live = qx >= 0.0 && qx <= res && qy >= 0.0 && qy <= res && qz >= 0.0 && qz <= res && qx < res && qy < res;
  }
  unsigned above = 0, below = 0;
  for (int f0 = 0; f0 < F; f0 += TILE) {
    const int nt = F - f0 < TILE ? F - f0 : TILE;
    __syncthreads();
    for (int k = threadIdx.x; k < nt * MESH_TC; k += blockDim.x) tile[k] = tc[(long)f0 * MESH_TC + k];
    __syncthreads();
    if (!live) continue;
    for (int j = 0; j < nt; ++j) {
      const double* c = tile + j * MESH_TC;
      const double adet = c[7];
      if (adet == 0.0) continue;
      const double y0 = qx - c[0], y1 = qy - c[1];
      const double u = (c[5] * y0 - c[3] * y1) * c[6];
      const double v = (-c[4] * y0 + c[2] * y1) * c[6];
      const double s = u + v;
      if (!(0.0 < u && u < adet && 0.0 < v && v < adet && 0.0 < s && s < adet)) continue;
      const double an = c[11];
      if (an == 0.0) continue;
      const double alpha = c[8] * (c[12] - qx) + c[9] * (c[13] - qy);
      const double depth = c[14] + alpha * c[10];
      const double zq = qz * an;
      if (depth >= zq) ++above;
      else if (depth < zq) ++below;
    }
  }
  if (i < N) out[i] = (unsigned char)(live && (above & 1u) && (below & 1u));
}

// ---- rigid transform of object point clouds along a trajectory (second half) ------------------------
// out[o, t, j] = R(o, t) pts[o, j] + tsl(o, t) with (tsl | rot6d) = traj[o, t]  - tslrot6d_to_transf_np + transf_point_array_np
// (dev_fn/transform/transform_np.py:169-175,36-53; rot6d_to_rotmat_np rotation_np.py:478-499: rows b1, b2, b1 x b2),
// as used by transf_merge_obj_pointcloud (compute_score_cr.py:122-137) and the SIV query points (compute_score_siv.py:146).
template <class T>
__global__ void transform_points_kernel(const T* __restrict__ traj, const T* __restrict__ pts, T* __restrict__ out, int nT, int P) {
  const int t = blockIdx.x, o = blockIdx.y;
  const T* tr = traj + ((long)o * nT + t) * 9;
  const T a0 = tr[3], a1 = tr[4], a2 = tr[5], c0 = tr[6], c1 = tr[7], c2 = tr[8];
  const T n1 = max(sqrt(a0 * a0 + a1 * a1 + a2 * a2), (T)1e-12);
  const T b1x = a0 / n1, b1y = a1 / n1, b1z = a2 / n1;
  const T dp = b1x * c0 + b1y * c1 + b1z * c2;
  T b2x = c0 - dp * b1x, b2y = c1 - dp * b1y, b2z = c2 - dp * b1z;
  const T n2 = max(sqrt(b2x * b2x + b2y * b2y + b2z * b2z), (T)1e-12);
  b2x /= n2; b2y /= n2; b2z /= n2;
  const T b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
  const T tx = tr[0], ty = tr[1], tz = tr[2];
  const T* pp = pts + (long)o * P * 3;
  T* op = out + (((long)o * nT + t) * P) * 3;
  for (int j = threadIdx.x; j < P; j += blockDim.x) {
    const T x = pp[j * 3 + 0], y = pp[j * 3 + 1], z = pp[j * 3 + 2];
    op[j * 3 + 0] = b1x * x + b1y * y + b1z * z + tx;
    op[j * 3 + 1] = b2x * x + b2y * y + b2z * z + ty;
    op[j * 3 + 2] = b3x * x + b3y * y + b3z * z + tz;
  }
}

// ---------------------------------------------------------------------------------------------
// Vertex normals of a triangle mesh sequence (the MANO hand: 778 vertices, 1 538 faces; segment_refine_model.py:131-133
// -> pytorch3d Meshes.verts_normals_packed, pytorch3d 0.7.2 structures/meshes.py _compute_vertex_normals):
//     n[v] = normalize( sum over the corners (f, c) with faces[f][c] == v of cross(x[next] - x[v], x[prev] - x[v]) ),
// next / prev = the face's following / preceding corner, i.e. every incident face contributes its area-weighted normal,
// computed with v as the pivot; normalize = x / max(|x|, 1e-6).  Gather form: the topology is fixed, so the host builds
// the incidence list once (CSR: off[V + 1], ent[2 * 3F] = (next, prev) vertex ids, ordered as index_add_ visits them on
// the CPU: corner 1 of all faces, then corner 2, then corner 0, faces ascending) and a thread sums its vertex's entries
// in that order - deterministic, no atomics.  No FMA contraction: the sums match the sequential CPU evaluation bit for bit.
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__global__ void vertex_normals_kernel(const float* __restrict__ verts, const int* __restrict__ off, const int* __restrict__ ent,
                                      float* __restrict__ out, long n_mesh, int V) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_mesh * V) return;
  const int v = (int)(idx % V);
  const float* x = verts + (idx / V) * (long)V * 3;
  const float px = x[3 * v], py = x[3 * v + 1], pz = x[3 * v + 2];
  float nx = 0.f, ny = 0.f, nz = 0.f;
  for (int e = off[v]; e < off[v + 1]; ++e) {
    const int a = ent[2 * e], b = ent[2 * e + 1];
    const float ax = x[3 * a] - px, ay = x[3 * a + 1] - py, az = x[3 * a + 2] - pz;
    const float bx = x[3 * b] - px, by = x[3 * b + 1] - py, bz = x[3 * b + 2] - pz;
    nx = nx + (ay * bz - az * by);
    ny = ny + (az * bx - ax * bz);
    nz = nz + (ax * by - ay * bx);
  }
  const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
  const float den = len > 1e-6f ? len : 1e-6f;
  out[idx * 3] = nx / den;
  out[idx * 3 + 1] = ny / den;
  out[idx * 3 + 2] = nz / den;
}
#pragma clang fp contract(fast)
