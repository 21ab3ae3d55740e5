// rtgs_common.hpp -- shared device/host helpers for the gfx950 ray-traced Gaussian-splat path.
//
// Math follows the reference's @ti.func code (paths relative to its src/rtgs/):
//   quaternion helpers  utils/quaternion.py:8-35, 84-121   (scalar-last x,y,z,w)
//   covariance          gaussian.py:86-102
// Nothing here is translated from the reference: it is the closed form of the same algebra.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtgs_hip.h"

namespace rtgs {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kTile = RTGS_TILE;   // macro-tile edge (pixels) = one workgroup
constexpr int kWaveTile = 8;       // one wave64 owns an 8x8 pixel tile
constexpr int kWavesX = kTile / kWaveTile;
constexpr int kWavesPerTile = kWavesX * kWavesX;
constexpr int kTraceBlock = kWavesPerTile * kWave;  // 256 threads for a 16x16 macro-tile
constexpr int kRecFloats = 16;     // per-frame projected record: 64 B = one cache line, one s_load_dwordx16
constexpr int kTailFloats = 8;     // second part of the record of the (rare) general-conic / near-surface Gaussians
constexpr float kInf = __builtin_huge_valf();

// utils/quaternion.py:8-23
template <typename T>
__host__ __device__ inline void quat_mul(const T* p, const T* q, T* out) {
  const T w = p[3] * q[3] - (p[0] * q[0] + p[1] * q[1] + p[2] * q[2]);
  const T x = p[3] * q[0] + q[3] * p[0] + (p[1] * q[2] - p[2] * q[1]);
  const T y = p[3] * q[1] + q[3] * p[1] + (p[2] * q[0] - p[0] * q[2]);
  const T z = p[3] * q[2] + q[3] * p[2] + (p[0] * q[1] - p[1] * q[0]);
  out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}

// utils/quaternion.py:26-35
template <typename T>
__host__ __device__ inline void quat_conj(const T* q, T* out) {
  out[0] = -q[0]; out[1] = -q[1]; out[2] = -q[2]; out[3] = q[3];
}

// utils/quaternion.py:84-96 -- (q (v,0) q*).xyz, the two Hamilton products written out.
template <typename T>
__host__ __device__ inline void quat_rot_vec3(const T* q, const T* v, T* out) {
  T qv[4] = {v[0], v[1], v[2], T(0)};
  T qc[4], t[4], r[4];
  quat_conj(q, qc);
  quat_mul(qv, qc, t);
  quat_mul(q, t, r);
  out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

// utils/quaternion.py:99-121 -- column k = q e_k q*.  For a non-unit q this is |q|^2 times a
// rotation, exactly as the reference's formula gives.  Row-major m[3*row + col].
// Closed form of the double product: (w^2 - u.u) I + 2 u u^T + 2 w [u]x.
template <typename T>
__host__ __device__ inline void quat_to_mat3(const T* q, T* m) {
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T s = w * w - (x * x + y * y + z * z);
  m[0] = s + 2 * x * x;       m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = s + 2 * y * y;       m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = s + 2 * z * z;
}

// gaussian.py:86-102 + ti.math.inverse (gaussian.py:196,215): Sigma^-1 = R S^-2 R^T / |q|^8, because
// the reference's R is |q|^2 times a rotation.  fp64, from the fp32 quaternion and scale.
// m6 = xx, xy, xz, yy, yz, zz.
__host__ __device__ inline void cov_inverse6(const float* rot, const float* scale, double* m6) {
  const double q[4] = {rot[0], rot[1], rot[2], rot[3]};
  double R[9];
  quat_to_mat3(q, R);
  const double k2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double k8 = (k2 * k2) * (k2 * k2);
  const double i0 = 1.0 / ((double)scale[0] * scale[0] * k8), i1 = 1.0 / ((double)scale[1] * scale[1] * k8),
               i2 = 1.0 / ((double)scale[2] * scale[2] * k8);
  int o = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 3; ++j)
      m6[o++] = R[3 * i] * i0 * R[3 * j] + R[3 * i + 1] * i1 * R[3 * j + 1] + R[3 * i + 2] * i2 * R[3 * j + 2];
}

// Per-frame camera constants, computed on the host in fp64 from the fp32 RtgsCamera.
struct FrameCam {
  double o[3];        // camera position
  double rc[9];       // d_world = rc * d_cam (row-major), camera.py:52 via rot_vec3
  double inv_fx, inv_fy;
  int32_t width, height;
};

inline FrameCam make_frame_cam(const RtgsCamera& c) {
  FrameCam f;
  for (int k = 0; k < 3; ++k) f.o[k] = c.position[k];
  const double q[4] = {c.rotation[0], c.rotation[1], c.rotation[2], c.rotation[3]};
  quat_to_mat3(q, f.rc);
  f.inv_fx = 1.0 / static_cast<double>(c.fx);
  f.inv_fy = 1.0 / static_cast<double>(c.fy);
  f.width = c.width;
  f.height = c.height;
  return f;
}

}  // namespace rtgs
