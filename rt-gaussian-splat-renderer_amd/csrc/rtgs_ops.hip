// rtgs_ops.hip -- the small kernels either side of the trace: covariance precompute, camera ray
// field, batched Scene.hit, and the elementwise device mirrors of the reference's @ti.func
// primitives (used by the Python API objects and by the parity tests).
//
// Reference (paths relative to its src/rtgs/): gaussian.py:86-102 (cov), camera.py:31-71,
// scene.py:406-450 (criteria :433-437), gaussian.py:140-230, utils/quaternion.py:8-121,
// ray.py:43-52, ray_tracer.py:56-77.
#include "rtgs_common.hpp"

namespace rtgs {
namespace {

inline int check_launch() { return hipGetLastError() == hipSuccess ? RTGS_OK : RTGS_ELAUNCH; }
inline dim3 grid_for(size_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

// ---- Gaussian.cov + inverse ---------------------------------------------------------
// Sigma = R S S^T R^T (gaussian.py:100-101); Sigma^-1 = R S^-2 R^T / |q|^8 because the
// reference's R is |q|^2 times a rotation (utils/quaternion.py:99-121).  fp64 inside so the
// stored fp32 values are correctly rounded.
__global__ __launch_bounds__(256) void k_covariance(int32_t n, const float* __restrict__ rot,
                                                     const float* __restrict__ scale,
                                                     float* __restrict__ cov6,
                                                     float* __restrict__ cov_inv6) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const double q[4] = {rot[4 * g], rot[4 * g + 1], rot[4 * g + 2], rot[4 * g + 3]};
  const double s[3] = {scale[3 * g], scale[3 * g + 1], scale[3 * g + 2]};
  double R[9];
  quat_to_mat3(q, R);
  const double k2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double k8 = (k2 * k2) * (k2 * k2);
  const double s2[3] = {s[0] * s[0], s[1] * s[1], s[2] * s[2]};
  const double is2[3] = {1.0 / (s2[0] * k8), 1.0 / (s2[1] * k8), 1.0 / (s2[2] * k8)};
  int o = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = i; j < 3; ++j) {
      double c = 0.0, ci = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        c += R[3 * i + k] * s2[k] * R[3 * j + k];
        ci += R[3 * i + k] * is2[k] * R[3 * j + k];
      }
      if (cov6) cov6[6 * g + o] = (float)c;
      cov_inv6[6 * g + o] = (float)ci;
      ++o;
    }
  }
}

// ---- Camera.generate_ray_field (camera.py:57-71), literal fp32 ------------------------
__global__ __launch_bounds__(256) void k_generate_rays(RtgsCamera cam, float* __restrict__ rays) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= cam.width * cam.height) return;
  const int i = p / cam.height, j = p % cam.height;
  const float W = (float)cam.width, H = (float)cam.height;
  const float u = ((float)i + 0.5f) / W;
  const float v = ((float)j + 0.5f) / H;
  const float px = (W * u - 0.5f * W) / cam.fx;
  const float py = (H * v - 0.5f * H) / cam.fy;
  const float inv = 1.0f / sqrtf(px * px + py * py + 1.0f);
  const float dc[3] = {px * inv, py * inv, -inv};
  float dw[3];
  quat_rot_vec3(cam.rotation, dc, dw);
  float* r = rays + (size_t)p * RTGS_RAY_FLOATS;
  float4* r4 = reinterpret_cast<float4*>(r);
  r4[0] = make_float4(cam.position[0], cam.position[1], cam.position[2], dw[0]);
  r4[1] = make_float4(dw[1], dw[2], 0.f, kInf);  // ray.py:20-41: start 0, end inf
}

// ---- ray/ellipsoid quadratic, gaussian.py:203-230, fp64 inside -------------------------
struct HitD { double t1, t2; };

template <typename TM>
__device__ __forceinline__ HitD hit_quadratic(const double o[3], const double d[3], const float* mu,
                                              const TM* ci) {
  const double inf = __builtin_huge_val();
  const double mx = o[0] - (double)mu[0], my = o[1] - (double)mu[1], mz = o[2] - (double)mu[2];
  const double Mxx = ci[0], Mxy = ci[1], Mxz = ci[2], Myy = ci[3], Myz = ci[4], Mzz = ci[5];
  const double Md0 = Mxx * d[0] + Mxy * d[1] + Mxz * d[2];
  const double Md1 = Mxy * d[0] + Myy * d[1] + Myz * d[2];
  const double Md2 = Mxz * d[0] + Myz * d[1] + Mzz * d[2];
  const double Mm0 = Mxx * mx + Mxy * my + Mxz * mz;
  const double Mm1 = Mxy * mx + Myy * my + Myz * mz;
  const double Mm2 = Mxz * mx + Myz * my + Mzz * mz;
  const double A = d[0] * Md0 + d[1] * Md1 + d[2] * Md2;
  const double B = 2.0 * (d[0] * Mm0 + d[1] * Mm1 + d[2] * Mm2);
  const double C = mx * Mm0 + my * Mm1 + mz * Mm2 - 3.0;  // BOUNDING_THRESHOLD, gaussian.py:13
  const double delta = B * B - 4.0 * A * C;
  HitD h{inf, inf};
  if (delta > 0.0) {
    const double sq = sqrt(delta);
    h.t1 = (-B - sq) / (2.0 * A);
    h.t2 = (-B + sq) / (2.0 * A);
  } else if (delta == 0.0) {
    h.t1 = -B / (2.0 * A);
  }
  return h;
}

// ---- Scene.hit for a batch of arbitrary rays (scene.py:406-450) ------------------------
// One thread per ray; the workgroup streams (position, rotation, scale) = 40 B/Gaussian, builds
// each slab's inverse covariances in fp64 and keeps the slab in LDS.  Brute force: the BVH of the reference only accelerates this query.
constexpr int kHitBlock = 256;
struct Closest { int idx; double t1, t2; };

// Every thread of the workgroup must call this (barriers inside); `live` rays take part.
__device__ __forceinline__ Closest closest_hit_block(int32_t n, const float* __restrict__ pos,
                                                     const float* __restrict__ rot,
                                                     const float* __restrict__ scale, bool live,
                                                     const double o[3], const double d[3], float start,
                                                     float end, float* s_mu, double* s_m) {
  const double inf = __builtin_huge_val();
  Closest c{-1, inf, inf};
  for (int base = 0; base < n; base += kHitBlock) {
    const int cnt = min(kHitBlock, n - base);
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
      const int g = base + threadIdx.x;
      float* s = s_mu + threadIdx.x * 3;
      s[0] = pos[3 * g]; s[1] = pos[3 * g + 1]; s[2] = pos[3 * g + 2];
      cov_inverse6(rot + 4 * (size_t)g, scale + 3 * (size_t)g, s_m + threadIdx.x * 6);
    }
    __syncthreads();
    if (live) {
      for (int k = 0; k < cnt; ++k) {
        const HitD h = hit_quadratic(o, d, s_mu + k * 3, s_m + k * 6);
        // scene.py:433-436: start < t1 < end, strictly smaller than the best so far.  The window
        // is tested on t1 ROUNDED TO FP32, the precision ray.start has: sample_step stores
        // start = t1 + 1e-8 = t1 in fp32 (ray_tracer.py:99-102), and the strict '>' must then
        // exclude that same Gaussian on the next peel, as it does in the all-fp32 reference.
        const float t1f = (float)h.t1;
        if (t1f < end && t1f > start && h.t1 < c.t1) {
          c.t1 = h.t1; c.t2 = h.t2; c.idx = base + k;
        }
      }
    }
  }
  return c;
}

// ---- acceleration structure for arbitrary-ray queries -------------------------------------
// The reference answers Scene.hit through a binned-SAH BVH over loose per-Gaussian boxes
// (scene.py:162-450, gaussian.py:104-138).  Here: the scene is stored in Morton order, so 64
// consecutive Gaussians (a slab) are spatially compact; the structure is, per Gaussian, the fp64
// inverse covariance (built ONCE per scene instead of once per workgroup per query), per slab the
// exact world-space box of its members' ellipsoids {d^T Sigma^-1 d <= 3} (half-extent sqrt(3 Sigma_ii)),
// and per 16 slabs the union of those.  A wave walks groups -> slabs -> Gaussians in index
// order; a box is skipped when no lane's ray can have a hit in it inside (start, min(end, best t1)).
// Same candidates' arithmetic as the brute-force kernel, same lowest-index tie rule: same answers.
constexpr int kAccGroup = 16;   // slabs per group box
constexpr int kBoxFloats = 8;   // lo xyz, hi xyz, 2 pad: one 32-B scalar load

struct AccelView {
  const double* minv;      // [nn][6]
  const float* slab_box;   // [nslabs][8]
  const float* group_box;  // [ngroups][8]
  const float* rad2;       // [nn] squared radius of the ball about the centre that holds the ellipsoid
  int32_t nslabs, ngroups;
};

__host__ __device__ inline size_t accel_nn(int32_t n) { return ((size_t)n + 63) & ~(size_t)63; }

__global__ __launch_bounds__(256) void k_accel_build(int32_t n, const float* __restrict__ pos,
                                                      const float* __restrict__ rot,
                                                      const float* __restrict__ scale,
                                                      double* __restrict__ minv, float* __restrict__ slab_box,
                                                      float* __restrict__ rad2) {
  const int g0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = g0 < n;
  const int g = live ? g0 : n - 1;
  double m6[6];
  cov_inverse6(rot + 4 * (size_t)g, scale + 3 * (size_t)g, m6);
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  if (live) {
#pragma unroll
    for (int k = 0; k < 6; ++k) minv[6 * (size_t)g + k] = m6[k];
    // Sigma_ii = sum_k R_ik^2 s_k^2 with the reference's R = |q|^2 x rotation (gaussian.py:100-101)
    const double q[4] = {rot[4 * g], rot[4 * g + 1], rot[4 * g + 2], rot[4 * g + 3]};
    double R[9];
    quat_to_mat3(q, R);
    const double s2[3] = {(double)scale[3 * g] * scale[3 * g], (double)scale[3 * g + 1] * scale[3 * g + 1],
                          (double)scale[3 * g + 2] * scale[3 * g + 2]};
    // {m : m^T Sigma^-1 m <= 3} has semi-axes sqrt(3) |q|^2 |s_k| (R is |q|^2 x a rotation): it lies
    // in the ball of that radius for the largest |s_k|.  Rounded up; NaN stays NaN (never rejects).
    const double k2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    rad2[g] = __double2float_ru(3.0 * k2 * k2 * fmax(s2[0], fmax(s2[1], s2[2])) * (1.0 + 1e-6));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double sii = R[3 * i] * R[3 * i] * s2[0] + R[3 * i + 1] * R[3 * i + 1] * s2[1] + R[3 * i + 2] * R[3 * i + 2] * s2[2];
      const double h = sqrt(3.0 * sii) * (1.0 + 1e-9);
      lo[i] = __double2float_rd((double)pos[3 * g + i] - h);   // rounded outwards; NaN stays NaN: ignored below
      hi[i] = __double2float_ru((double)pos[3 * g + i] + h);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      lo[i] = fminf(lo[i], __shfl_xor(lo[i], off));
      hi[i] = fmaxf(hi[i], __shfl_xor(hi[i], off));
    }
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && live) {
    float* b = slab_box + (size_t)(g0 / kWave) * kBoxFloats;
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2]; b[6] = 0.f; b[7] = 0.f;
  }
}

__global__ __launch_bounds__(256) void k_accel_groups(int32_t nslabs, const float* __restrict__ slab_box,
                                                       float* __restrict__ group_box) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u * kAccGroup >= nslabs) return;
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  for (int k = 0; k < kAccGroup; ++k) {
    const int sl = u * kAccGroup + k;
    if (sl >= nslabs) break;
    const float* b = slab_box + (size_t)sl * kBoxFloats;
#pragma unroll
    for (int i = 0; i < 3; ++i) { lo[i] = fminf(lo[i], b[i]); hi[i] = fmaxf(hi[i], b[3 + i]); }
  }
  float* o = group_box + (size_t)u * kBoxFloats;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2]; o[6] = 0.f; o[7] = 0.f;
}

// Can the ray (o + t d, fp32 as stored) have a point inside the box for some t in [t_lo, t_hi]?
// Conservative: the slab distances are padded by 1e-6 relative.  A zero direction component
// constrains nothing if the origin lies between the two planes, everything otherwise.
// tn_out: the (padded) parameter at which the ray enters the box.
__device__ __forceinline__ bool box_may_hit(const float* __restrict__ b, const float o[3], const float d[3],
                                            const float inv[3], float t_lo, float t_hi, float* tn_out = nullptr) {
  float tn = -kInf, tf = kInf;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float t0 = (b[i] - o[i]) * inv[i], t1 = (b[3 + i] - o[i]) * inv[i];
    float a = fminf(t0, t1), c = fmaxf(t0, t1);
    if (d[i] == 0.f) {
      const bool in = o[i] >= b[i] && o[i] <= b[3 + i];
      a = in ? -kInf : kInf;
      c = in ? kInf : -kInf;
    }
    tn = fmaxf(tn, a);
    tf = fminf(tf, c);
  }
  tn -= fabsf(tn) * 1e-6f;
  tf += fabsf(tf) * 1e-6f;
  if (tn_out) *tn_out = tn;
  return tn <= tf && tf >= t_lo && tn <= t_hi;
}

// wave-wide minimum of an unsigned value (xor butterfly)
__device__ __forceinline__ unsigned wave_umin_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, off));
  return v;
}

__device__ __forceinline__ Closest closest_hit_accel(const AccelView av, int32_t n, const float* __restrict__ pos,
                                                     bool live, const double o[3], const double d[3],
                                                     float start, float end) {
  const double inf = __builtin_huge_val();
  Closest c{-1, inf, inf};
  const float fo[3] = {(float)o[0], (float)o[1], (float)o[2]};   // exact: rays are stored in fp32
  const float fd[3] = {(float)d[0], (float)d[1], (float)d[2]};
  const float inv[3] = {1.0f / fd[0], 1.0f / fd[1], 1.0f / fd[2]};
  const int lane = threadIdx.x & (kWave - 1);
  const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  float t_hi = end;   // min(end, best t1 so far), rounded up
  // Groups are visited NEAREST FIRST within each run of 64 (by the smallest distance at which any
  // ray of the wave enters the group's box), so that the first hits shrink every ray's window
  // before the far groups are looked at; index order made a ray walk every box along its whole
  // length.  The answer does not depend on the order: the closest hit is the minimum of (t1, index).
  // (The 16 slabs of a group nearest first as well: c3 1.24 -> 1.17 ms, c5 5.30 -> 5.40: not kept.)
  constexpr unsigned kSkip = 0xFFFFFFFFu;
  auto group_key = [&](int u) -> unsigned {
    float tn;
    const bool hu = live && box_may_hit(av.group_box + (size_t)u * kBoxFloats, fo, fd, inv, start, t_hi, &tn);
    // entry distances below `start` order alike; non-negative floats order like their bits
    return hu ? __float_as_uint(fmaxf(fmaxf(tn, start), 0.f)) : kSkip;
  };
  auto visit_run = [&](int base) {
    const int cnt = min(kWave, av.ngroups - base);
    unsigned mykey = kSkip;   // lane j: key of group base + j
    for (int j = 0; j < cnt; ++j) {
      const unsigned kmin = wave_umin_u32(group_key(base + j));
      if (lane == j) mykey = kmin;
    }
    while (true) {
      const unsigned kmin = wave_umin_u32(mykey);
      if (kmin == kSkip) break;
      const int j = __builtin_ctzll(__ballot(mykey == kmin));
      if (lane == j) mykey = kSkip;
      const int u = base + j;
      const bool hu = live && box_may_hit(av.group_box + (size_t)u * kBoxFloats, fo, fd, inv, start, t_hi);
      if (!__any(hu)) continue;
      const int s_end = min(av.nslabs, (u + 1) * kAccGroup);
      for (int sl = u * kAccGroup; sl < s_end; ++sl) {
        const bool hs = hu && box_may_hit(av.slab_box + (size_t)sl * kBoxFloats, fo, fd, inv, start, t_hi);
        if (!__any(hs)) continue;
        const int g_end = min(n, (sl + 1) * kWave);
        for (int g = sl * kWave; g < g_end; ++g) {
          if (hs) {
            // the ray's line must pass through the Gaussian's bounding ball: |m x d|^2 <= r^2 |d|^2 with
            // m = o - mu -- a dozen fp64 operations that spare most candidates of a slab the quadratic.
            // Conservative (radius rounded up, slack for the cancellation); NaNs fall through.
            const float* mu = pos + 3 * (size_t)g;
            const double mx = o[0] - (double)mu[0], my = o[1] - (double)mu[1], mz = o[2] - (double)mu[2];
            const double md = mx * d[0] + my * d[1] + mz * d[2], mm = mx * mx + my * my + mz * mz;
            if (mm * dd - md * md > (double)av.rad2[g] * dd * (1.0 + 1e-6) + 1e-13 * mm * dd) continue;
            const HitD h = hit_quadratic(o, d, mu, av.minv + 6 * (size_t)g);
            const float t1f = (float)h.t1;   // the window is tested in fp32: see closest_hit_block
            // (t1, index) lexicographic: what increasing-index order with a strict `<` gives
            if (t1f < end && t1f > start && (h.t1 < c.t1 || (h.t1 == c.t1 && g < c.idx))) {
              c.t1 = h.t1; c.t2 = h.t2; c.idx = g;
              t_hi = fminf(t_hi, __double2float_ru(h.t1));
            }
          }
        }
      }
    }
  };
  const int nruns = (av.ngroups + kWave - 1) / kWave;
  if (nruns > 1 && nruns <= kWave) {
    // the runs themselves nearest first (lane r keeps run r's key: up to 64 runs = 4 M Gaussians;
    // beyond that the runs are taken in index order)
    unsigned rkey = kSkip;
    for (int r = 0; r < nruns; ++r) {
      unsigned k = kSkip;
      const int u_end = min(av.ngroups, (r + 1) * kWave);
      for (int u = r * kWave; u < u_end; ++u) k = min(k, group_key(u));
      k = wave_umin_u32(k);
      if (lane == r) rkey = k;
    }
    while (true) {
      const unsigned kmin = wave_umin_u32(rkey);
      if (kmin == kSkip) break;
      const int r = __builtin_ctzll(__ballot(rkey == kmin));
      if (lane == r) rkey = kSkip;
      visit_run(r * kWave);
    }
  } else {
    for (int base = 0; base < av.ngroups; base += kWave) visit_run(base);
  }
  return c;
}

__global__ __launch_bounds__(kHitBlock) void k_closest_hit(int32_t n, const float* __restrict__ pos,
                                                            const float* __restrict__ rot,
                                                            const float* __restrict__ scale,
                                                            const AccelView av, int32_t n_rays,
                                                            const float* __restrict__ rays,
                                                            int32_t* __restrict__ hit_idx,
                                                            float* __restrict__ hit_t) {
  __shared__ float s_mu[kHitBlock * 3];
  __shared__ double s_m[kHitBlock * 6];   // inverse covariances of the slab, built here in fp64
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = r < n_rays;
  double o[3] = {0, 0, 0}, d[3] = {0, 1, 0};
  float start = 0.f, end = kInf;
  if (live) {
    const float* rp = rays + (size_t)r * RTGS_RAY_FLOATS;
    o[0] = rp[0]; o[1] = rp[1]; o[2] = rp[2];
    d[0] = rp[3]; d[1] = rp[4]; d[2] = rp[5];
    start = rp[6]; end = rp[7];
  }
  const Closest c = av.minv ? closest_hit_accel(av, n, pos, live, o, d, start, end)
                            : closest_hit_block(n, pos, rot, scale, live, o, d, start, end, s_mu, s_m);
  if (live) {
    hit_idx[r] = c.idx;
    hit_t[2 * r] = (float)c.t1;
    hit_t[2 * r + 1] = (float)c.t2;
  }
}

__device__ __forceinline__ void gaussian_eval_d(const double p[3], const double dir[3], const float* mu,
                                                const float* rot, const float* scale, const float* colop,
                                                const float* sh, double out[4]);

// ---- RayTracer.sample_step for an ARBITRARY ray field (ray_tracer.py:79-104, literally) -------
// Scene.hit by brute force, Gaussian.eval at the chord midpoint, accumulate, advance ray.start.
// The camera-ray fast path is k_trace (csrc/rtgs_trace.hip); this one serves ray fields the
// caller has edited, which k_trace cannot assume anything about.
__global__ __launch_bounds__(kHitBlock) void k_sample_step_rays(int32_t n, const float* __restrict__ pos,
                                                                 const float* __restrict__ rot,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ colop,
                                                                 const float* __restrict__ sh,
                                                                 const AccelView av,
                                                                 int32_t n_rays, float* __restrict__ rays,
                                                                 float* __restrict__ sample,
                                                                 float* __restrict__ att) {
  __shared__ float s_mu[kHitBlock * 3];
  __shared__ double s_m[kHitBlock * 6];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = r < n_rays;
  double o[3] = {0, 0, 0}, d[3] = {0, 1, 0};
  float start = 0.f, end = kInf;
  float* rp = rays + (size_t)r * RTGS_RAY_FLOATS;
  if (live) {
    o[0] = rp[0]; o[1] = rp[1]; o[2] = rp[2];
    d[0] = rp[3]; d[1] = rp[4]; d[2] = rp[5];
    start = rp[6]; end = rp[7];
  }
  const Closest c = av.minv ? closest_hit_accel(av, n, pos, live, o, d, start, end)
                            : closest_hit_block(n, pos, rot, scale, live, o, d, start, end, s_mu, s_m);
  if (!live) return;
  if (c.idx < 0) {
    rp[6] = kInf;   // ray_tracer.py:103-104
    return;
  }
  const double tm = (c.t1 + c.t2) / 2.0;   // ray_tracer.py:90-91
  const double p[3] = {o[0] + tm * d[0], o[1] + tm * d[1], o[2] + tm * d[2]};
  const size_t g = (size_t)c.idx;
  double e[4];
  gaussian_eval_d(p, d, pos + 3 * g, rot + 4 * g, scale + 3 * g, colop + 4 * g,
                  sh ? sh + g * RTGS_SH_COEFFS * 3 : nullptr, e);
  const float a = att[r];
  const float alpha = (float)e[3];
  sample[3 * (size_t)r + 0] += a * alpha * (float)e[0];   // ray_tracer.py:95-97
  sample[3 * (size_t)r + 1] += a * alpha * (float)e[1];
  sample[3 * (size_t)r + 2] += a * alpha * (float)e[2];
  att[r] = a * (1.0f - alpha);
  rp[6] = (float)c.t1 + 1e-8f;   // ray_tracer.py:99-102 (absorbed in fp32 for t1 > ~0.17)
}

// ---- Gaussian.hit, elementwise ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_gaussian_hit(int32_t n, const float* __restrict__ rays,
                                                       const float* __restrict__ pos,
                                                       const float* __restrict__ rot,
                                                       const float* __restrict__ scale,
                                                       float* __restrict__ hit_t) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float* rp = rays + (size_t)k * RTGS_RAY_FLOATS;
  const double o[3] = {rp[0], rp[1], rp[2]};
  const double d[3] = {rp[3], rp[4], rp[5]};
  double m6[6];
  cov_inverse6(rot + 4 * (size_t)k, scale + 3 * (size_t)k, m6);
  const HitD h = hit_quadratic(o, d, pos + 3 * (size_t)k, m6);
  hit_t[2 * k] = (float)h.t1;
  hit_t[2 * k + 1] = (float)h.t2;
}

// ---- Gaussian.eval + eval_sh (gaussian.py:140-201), fp64 inside -----------------------------
// out = (rgb, alpha) of Gaussian (mu, rot, scale, colop[, sh]) at point p seen along dir.
__device__ __forceinline__ void gaussian_eval_d(const double p[3], const double dir[3], const float* mu,
                                                const float* rot, const float* scale, const float* colop,
                                                const float* sh, double out[4]) {
  const double x = p[0] - (double)mu[0], y = p[1] - (double)mu[1], z = p[2] - (double)mu[2];
  double c[6];
  cov_inverse6(rot, scale, c);
  const double q = x * (c[0] * x + c[1] * y + c[2] * z) + y * (c[1] * x + c[3] * y + c[4] * z) +
                   z * (c[2] * x + c[4] * y + c[5] * z);
  const double rho = exp(-q);  // no 1/2 factor: gaussian.py:197
  double rgb[3] = {colop[0], colop[1], colop[2]};
  if (sh) {
    double dx = dir[0], dy = dir[1], dz = dir[2];
    const double il = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);  // gaussian.py:200
    dx *= il; dy *= il; dz *= il;
    const double c0 = sqrt(3.0 / M_PI), c1 = sqrt(15.0 / M_PI), c2 = sqrt(5.0 / M_PI),
                 c3 = sqrt(35.0 / (2.0 * M_PI)), c4 = sqrt(105.0 / M_PI),
                 c5 = sqrt(21.0 / (2.0 * M_PI)), c6 = sqrt(7.0 / M_PI);
    const double Y[RTGS_SH_COEFFS] = {
        0.5 * c0 * dy, 0.5 * c0 * dz, 0.5 * c0 * dx, 0.5 * c1 * dx * dy, 0.5 * c1 * dy * dz,
        0.25 * c2 * (3 * dz * dz - 1), 0.5 * c1 * dx * dz, 0.25 * c1 * (dx * dx - dy * dy),
        0.25 * c3 * dy * (3 * dx * dx - dy * dy), 0.5 * c4 * dx * dy * dz,
        0.25 * c5 * dy * (5 * dz * dz - 1), 0.25 * c6 * (5 * dz * dz - 3 * dz) /* sic, :160 */,
        0.25 * c5 * dx * (5 * dz * dz - 1), 0.25 * c4 * (dx * dx - dy * dy) * dz,
        0.25 * c3 * dx * (dx * dx - 3 * dy * dy)};
    for (int b = 0; b < RTGS_SH_COEFFS; ++b) {
      rgb[0] += Y[b] * sh[3 * b]; rgb[1] += Y[b] * sh[3 * b + 1]; rgb[2] += Y[b] * sh[3 * b + 2];
    }
  }
  out[0] = rgb[0]; out[1] = rgb[1]; out[2] = rgb[2];
  out[3] = (double)colop[3] * rho;
}

__global__ __launch_bounds__(256) void k_gaussian_eval(int32_t n, const float* __restrict__ pos,
                                                        const float* __restrict__ dir,
                                                        const float* __restrict__ mu,
                                                        const float* __restrict__ rot,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ colop,
                                                        const float* __restrict__ sh,
                                                        float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double p[3] = {pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]};
  const double d[3] = {dir[3 * k], dir[3 * k + 1], dir[3 * k + 2]};
  double o[4];
  gaussian_eval_d(p, d, mu + 3 * (size_t)k, rot + 4 * (size_t)k, scale + 3 * (size_t)k,
                  colop + 4 * (size_t)k, sh ? sh + (size_t)k * RTGS_SH_COEFFS * 3 : nullptr, o);
  out[4 * k] = (float)o[0]; out[4 * k + 1] = (float)o[1]; out[4 * k + 2] = (float)o[2];
  out[4 * k + 3] = (float)o[3];
}

// ---- quaternion helpers, literal fp32 (utils/quaternion.py) -------------------------------
__global__ __launch_bounds__(256) void k_quat_op(int32_t op, int32_t n, const float* __restrict__ p,
                                                  const float* __restrict__ q,
                                                  float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float pp[4] = {p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]};
  if (op == 0) {
    const float qq[4] = {q[4 * k], q[4 * k + 1], q[4 * k + 2], q[4 * k + 3]};
    float r[4];
    quat_mul(pp, qq, r);
    for (int c = 0; c < 4; ++c) out[4 * k + c] = r[c];
  } else if (op == 1) {
    float r[4];
    quat_conj(pp, r);
    for (int c = 0; c < 4; ++c) out[4 * k + c] = r[c];
  } else if (op == 2) {
    const float v[3] = {q[4 * k], q[4 * k + 1], q[4 * k + 2]};
    float r[3];
    quat_rot_vec3(pp, v, r);
    for (int c = 0; c < 3; ++c) out[3 * k + c] = r[c];
  } else {
    // as_rotation_mat3: column c = q e_c q* via the same two products the reference runs
    for (int c = 0; c < 3; ++c) {
      float e[3] = {0.f, 0.f, 0.f};
      e[c] = 1.f;
      float r[3];
      quat_rot_vec3(pp, e, r);
      out[9 * k + 0 + c] = r[0]; out[9 * k + 3 + c] = r[1]; out[9 * k + 6 + c] = r[2];
    }
  }
}

__global__ __launch_bounds__(256) void k_ray_get(int32_t n, const float* __restrict__ rays,
                                                  const float* __restrict__ t,
                                                  float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float* r = rays + (size_t)k * RTGS_RAY_FLOATS;
  for (int c = 0; c < 3; ++c) out[3 * k + c] = r[c] + t[k] * r[3 + c];  // ray.py:52
}

__global__ __launch_bounds__(256) void k_disp(size_t n, const float* __restrict__ sample, float denom,
                                               float* __restrict__ disp) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) disp[k] = sample[k] / denom;  // ray_tracer.py:76-77
}

__global__ __launch_bounds__(256) void k_fill(float* __restrict__ dst, size_t n, float value) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) dst[k] = value;
}

}  // namespace
}  // namespace rtgs

using namespace rtgs;

extern "C" {

int rtgs_abi_version(void) { return RTGS_ABI_VERSION; }

const char* rtgs_strerror(int code) {
  switch (code) {
    case RTGS_OK: return "ok";
    case RTGS_EINVAL: return "invalid argument (null pointer, non-positive size or bad enum)";
    case RTGS_EDEPTH: return "depth < 1, or depth > RTGS_MAX_DEPTH without the per-pixel state buffers (attenuation_buf, ray_start)";
    case RTGS_EWORKSPACE: return "workspace smaller than rtgs_workspace_bytes()";
    case RTGS_ELAUNCH: return "HIP kernel launch failed";
    case RTGS_EDEVICE: return "no usable HIP device";
    default: return "unknown rtgs error";
  }
}

int rtgs_covariance(int32_t n, const float* rotation, const float* scale, float* cov6,
                    float* cov_inv6, rtgs_stream_t stream) {
  if (n <= 0 || !rotation || !scale || !cov_inv6) return RTGS_EINVAL;
  hipLaunchKernelGGL(k_covariance, grid_for(n, 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, rotation, scale, cov6, cov_inv6);
  return check_launch();
}

int rtgs_precompute(int32_t n, const float* position, const float* rotation, const float* scale,
                    float* sigma_inv6, rtgs_stream_t stream) {
  (void)position;   // the inverse covariance does not depend on it; kept for the SURVEY 8(b) signature
  return rtgs_covariance(n, rotation, scale, nullptr, sigma_inv6, stream);
}

int rtgs_generate_rays(const RtgsCamera* cam, float* rays, rtgs_stream_t stream) {
  if (!cam || !rays || cam->width <= 0 || cam->height <= 0 || !(cam->fx > 0.f) || !(cam->fy > 0.f))
    return RTGS_EINVAL;
  hipLaunchKernelGGL(k_generate_rays, grid_for((size_t)cam->width * cam->height, 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *cam, rays);
  return check_launch();
}

size_t rtgs_accel_bytes(int32_t n) {
  if (n <= 0) return 0;
  const size_t nn = accel_nn(n), nslabs = nn / kWave, ngroups = (nslabs + kAccGroup - 1) / kAccGroup;
  return nn * 6 * sizeof(double) + (nslabs + ngroups) * kBoxFloats * sizeof(float) + nn * sizeof(float);
}

static int accel_view(const RtgsScene* scene, const void* accel, size_t accel_bytes, AccelView* av) {
  *av = AccelView{nullptr, nullptr, nullptr, nullptr, 0, 0};
  if (!accel) return RTGS_OK;   // brute force
  if (accel_bytes < rtgs_accel_bytes(scene->n)) return RTGS_EWORKSPACE;
  if (reinterpret_cast<uintptr_t>(accel) & 15u) return RTGS_EINVAL;
  const size_t nn = accel_nn(scene->n);
  av->nslabs = (int32_t)(nn / kWave);
  av->ngroups = (av->nslabs + kAccGroup - 1) / kAccGroup;
  av->minv = static_cast<const double*>(accel);
  av->slab_box = reinterpret_cast<const float*>(av->minv + nn * 6);
  av->group_box = av->slab_box + (size_t)av->nslabs * kBoxFloats;
  av->rad2 = av->group_box + (size_t)av->ngroups * kBoxFloats;
  return RTGS_OK;
}

int rtgs_accel_build(const RtgsScene* scene, void* accel, size_t accel_bytes, rtgs_stream_t stream) {
  if (!scene || scene->n <= 0 || !scene->position || !scene->rotation || !scene->scale || !accel) return RTGS_EINVAL;
  AccelView av;
  if (int rc = accel_view(scene, accel, accel_bytes, &av)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_accel_build, grid_for(scene->n, 256), dim3(256), 0, s, scene->n, scene->position,
                     scene->rotation, scene->scale, const_cast<double*>(av.minv), const_cast<float*>(av.slab_box),
                     const_cast<float*>(av.rad2));
  hipLaunchKernelGGL(k_accel_groups, grid_for(av.ngroups, 256), dim3(256), 0, s, av.nslabs, av.slab_box,
                     const_cast<float*>(av.group_box));
  return check_launch();
}

int rtgs_closest_hit_accel(const RtgsScene* scene, const void* accel, size_t accel_bytes, int32_t n_rays,
                           const float* rays, int32_t* hit_idx, float* hit_t, rtgs_stream_t stream) {
  if (!scene || scene->n <= 0 || !scene->position || !scene->rotation || !scene->scale) return RTGS_EINVAL;
  if (n_rays <= 0 || !rays || !hit_idx || !hit_t) return RTGS_EINVAL;
  AccelView av;
  if (int rc = accel_view(scene, accel, accel_bytes, &av)) return rc;
  hipLaunchKernelGGL(k_closest_hit, grid_for(n_rays, kHitBlock), dim3(kHitBlock), 0,
                     static_cast<hipStream_t>(stream), scene->n, scene->position, scene->rotation,
                     scene->scale, av, n_rays, rays, hit_idx, hit_t);
  return check_launch();
}

int rtgs_closest_hit(const RtgsScene* scene, int32_t n_rays, const float* rays, int32_t* hit_idx,
                     float* hit_t, rtgs_stream_t stream) {
  return rtgs_closest_hit_accel(scene, nullptr, 0, n_rays, rays, hit_idx, hit_t, stream);
}

int rtgs_sample_step_rays_accel(const RtgsScene* scene, const void* accel, size_t accel_bytes, int32_t n_rays,
                                float* rays, float* sample_buf, float* attenuation_buf, rtgs_stream_t stream) {
  if (!scene || scene->n <= 0 || !scene->position || !scene->rotation || !scene->scale ||
      !scene->color_opacity || n_rays < 0 || !rays || !sample_buf || !attenuation_buf)
    return RTGS_EINVAL;
  if (n_rays == 0) return RTGS_OK;
  AccelView av;
  if (int rc = accel_view(scene, accel, accel_bytes, &av)) return rc;
  hipLaunchKernelGGL(k_sample_step_rays, grid_for(n_rays, kHitBlock), dim3(kHitBlock), 0,
                     static_cast<hipStream_t>(stream), scene->n, scene->position, scene->rotation,
                     scene->scale, scene->color_opacity, scene->sh, av, n_rays, rays, sample_buf,
                     attenuation_buf);
  return check_launch();
}

int rtgs_sample_step_rays(const RtgsScene* scene, int32_t n_rays, float* rays, float* sample_buf,
                          float* attenuation_buf, rtgs_stream_t stream) {
  return rtgs_sample_step_rays_accel(scene, nullptr, 0, n_rays, rays, sample_buf, attenuation_buf, stream);
}

int rtgs_gaussian_hit(int32_t n, const float* rays, const float* position, const float* rotation,
                      const float* scale, float* hit_t, rtgs_stream_t stream) {
  if (n <= 0 || !rays || !position || !rotation || !scale || !hit_t) return RTGS_EINVAL;
  hipLaunchKernelGGL(k_gaussian_hit, grid_for(n, 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, rays, position, rotation, scale, hit_t);
  return check_launch();
}

int rtgs_gaussian_eval(int32_t n, const float* pos, const float* dir, const float* position,
                       const float* rotation, const float* scale, const float* color_opacity,
                       const float* sh, float* out, rtgs_stream_t stream) {
  if (n <= 0 || !pos || !dir || !position || !rotation || !scale || !color_opacity || !out) return RTGS_EINVAL;
  hipLaunchKernelGGL(k_gaussian_eval, grid_for(n, 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, pos, dir, position, rotation, scale, color_opacity, sh, out);
  return check_launch();
}

int rtgs_quat_op(int32_t op, int32_t n, const float* p, const float* q, float* out,
                 rtgs_stream_t stream) {
  if (op < 0 || op > 3 || n <= 0 || !p || !out) return RTGS_EINVAL;
  if ((op == 0 || op == 2) && !q) return RTGS_EINVAL;
  hipLaunchKernelGGL(k_quat_op, grid_for(n, 256), dim3(256), 0, static_cast<hipStream_t>(stream), op, n,
                     p, q, out);
  return check_launch();
}

int rtgs_ray_get(int32_t n, const float* rays, const float* t, float* out, rtgs_stream_t stream) {
  if (n <= 0 || !rays || !t || !out) return RTGS_EINVAL;
  hipLaunchKernelGGL(k_ray_get, grid_for(n, 256), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                     rays, t, out);
  return check_launch();
}

int rtgs_disp_buffer(int32_t n_pixels, const float* sample_buf, int32_t num_samples, int32_t num_steps,
                     int32_t num_depth, float* disp_buf, rtgs_stream_t stream) {
  if (n_pixels <= 0 || !sample_buf || !disp_buf || num_depth <= 0) return RTGS_EINVAL;
  // ray_tracer.py:76-77: num_samples + float(num_steps) / float(num_depth)
  const float denom = (float)num_samples + (float)num_steps / (float)num_depth;
  const size_t n = (size_t)n_pixels * 3;
  hipLaunchKernelGGL(k_disp, grid_for(n, 256), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                     sample_buf, denom, disp_buf);
  return check_launch();
}

int rtgs_fill(float* dst, size_t n, float value, rtgs_stream_t stream) {
  if (!dst || n == 0) return RTGS_EINVAL;
  hipLaunchKernelGGL(k_fill, grid_for(n, 256), dim3(256), 0, static_cast<hipStream_t>(stream), dst, n,
                     value);
  return check_launch();
}

}  // extern "C"
