/*
 * rtgs_oracle.c -- CPU oracle, plain C (fp64, OpenMP over pixels).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this.  The product (rtgs package + librtgs_hip.so) never does.
 *
 * A second, independently written restatement of the reference renderer
 * (fangjunzhou/rt-gaussian-splat-renderer @ 2025-06-29, paths relative to its src/rtgs/), used to
 * cross-check oracle/rtgs_ref.py and to evaluate large pixel samples quickly.  It follows the
 * reference literally where rtgs_ref.render_topk uses a closed form:
 *   - rotation matrix from the two quaternion products per basis vector, utils/quaternion.py:84-121
 *   - Sigma = R S S^T R^T and its inverse by adjugate/determinant, gaussian.py:86-102, :196, :215
 *   - ray generation, camera.py:31-71
 *   - A, B, C, delta, t1, t2, gaussian.py:215-230
 *   - alpha = opacity * exp(-(x-mu)^T Sigma^-1 (x-mu)) at the chord midpoint x = o + (t1+t2)/2 d,
 *     ray_tracer.py:88-91 + gaussian.py:194-198; colour + SH, gaussian.py:140-181, 199-200
 *   - closest-hit order and compositing, scene.py:433-437, ray_tracer.py:93-104, restated as the
 *     `depth` smallest t1 > 0 composited front to back; an exact repeat
 *     of the previous t1 is dropped, as the reference's strict `t1 > ray.start` does.
 * PARITY UNPINNED for these functions (no reference test pins a value; taichi is absent here):
 * see the header of rtgs_ref.py.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SH_COEFFS 15

static void quat_mul(const double* p, const double* q, double* o) { /* utils/quaternion.py:8-23 */
  double w = p[3] * q[3] - (p[0] * q[0] + p[1] * q[1] + p[2] * q[2]);
  double x = p[3] * q[0] + q[3] * p[0] + (p[1] * q[2] - p[2] * q[1]);
  double y = p[3] * q[1] + q[3] * p[1] + (p[2] * q[0] - p[0] * q[2]);
  double z = p[3] * q[2] + q[3] * p[2] + (p[0] * q[1] - p[1] * q[0]);
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

static void rot_vec3(const double* q, const double* v, double* o) { /* utils/quaternion.py:84-96 */
  double qv[4] = {v[0], v[1], v[2], 0.0}, qc[4] = {-q[0], -q[1], -q[2], q[3]}, t[4], r[4];
  quat_mul(qv, qc, t);
  quat_mul(q, t, r);
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}

static void cov_inverse(const double* q, const double* s, double* M /*3x3 row-major*/) {
  double R[9], RS[9], S[9];
  for (int c = 0; c < 3; ++c) { /* utils/quaternion.py:99-121 */
    double e[3] = {0, 0, 0}, r[3];
    e[c] = 1.0;
    rot_vec3(q, e, r);
    R[0 + c] = r[0]; R[3 + c] = r[1]; R[6 + c] = r[2];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) RS[3 * i + j] = R[3 * i + j] * s[j] * s[j]; /* R S S^T */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += RS[3 * i + k] * R[3 * j + k];
      S[3 * i + j] = a; /* gaussian.py:100-101 */
    }
  /* adjugate / determinant (ti.math.inverse) */
  double c00 = S[4] * S[8] - S[5] * S[7], c01 = -(S[3] * S[8] - S[5] * S[6]), c02 = S[3] * S[7] - S[4] * S[6];
  double det = S[0] * c00 + S[1] * c01 + S[2] * c02;
  M[0] = c00 / det; M[3] = c01 / det; M[6] = c02 / det;
  M[1] = -(S[1] * S[8] - S[2] * S[7]) / det; M[4] = (S[0] * S[8] - S[2] * S[6]) / det;
  M[7] = -(S[0] * S[7] - S[1] * S[6]) / det;
  M[2] = (S[1] * S[5] - S[2] * S[4]) / det; M[5] = -(S[0] * S[5] - S[2] * S[3]) / det;
  M[8] = (S[0] * S[4] - S[1] * S[3]) / det;
}

static double quadform(const double* M, const double* a, const double* b) {
  double r = 0;
  for (int i = 0; i < 3; ++i) r += a[i] * (M[3 * i] * b[0] + M[3 * i + 1] * b[1] + M[3 * i + 2] * b[2]);
  return r;
}

static void sh_basis(const double* d, double* Y) { /* gaussian.py:17-23, 149-163 */
  const double c0 = sqrt(3 / M_PI), c1 = sqrt(15 / M_PI), c2 = sqrt(5 / M_PI), c3 = sqrt(35 / (2 * M_PI)),
               c4 = sqrt(105 / M_PI), c5 = sqrt(21 / (2 * M_PI)), c6 = sqrt(7 / M_PI);
  const double x = d[0], y = d[1], z = d[2];
  Y[0] = 0.5 * c0 * y; Y[1] = 0.5 * c0 * z; Y[2] = 0.5 * c0 * x;
  Y[3] = 0.5 * c1 * x * y; Y[4] = 0.5 * c1 * y * z; Y[5] = 0.25 * c2 * (3 * z * z - 1);
  Y[6] = 0.5 * c1 * x * z; Y[7] = 0.25 * c1 * (x * x - y * y);
  Y[8] = 0.25 * c3 * y * (3 * x * x - y * y); Y[9] = 0.5 * c4 * x * y * z;
  Y[10] = 0.25 * c5 * y * (5 * z * z - 1); Y[11] = 0.25 * c6 * (5 * z * z - 3 * z); /* sic :160 */
  Y[12] = 0.25 * c5 * x * (5 * z * z - 1); Y[13] = 0.25 * c4 * (x * x - y * y) * z;
  Y[14] = 0.25 * c3 * x * (x * x - 3 * y * y);
}

/* Returns 0 on success.  All array arguments are host pointers.
 * sample[n_pix][3], attenuation[n_pix], start[n_pix] (t1 of the last peel or +inf), hits[n_pix],
 * margin_q / margin_t [n_pix]: distance of the pixel from a discontinuity of the image function
 * (see rtgs_ref.render_topk).
 * t_cutoff > 0 (an extension of the build; the reference has no early-out, ray_tracer.py:79-104):
 * a ray whose attenuation is below t_cutoff composites nothing more; margin_c [n_pix] (may be
 * NULL) = the closest an attenuation on the way came to t_cutoff, relative. */
int rtgs_oracle_render(int n, const float* pos, const float* rot, const float* scale, const float* color,
                       const float* opacity, const float* sh /* [n][15][3] or NULL */, const float* cam_pos,
                       const float* cam_rot, int W, int H, float fx, float fy, int depth, int n_pix,
                       const int* pixels, double* sample, double* attenuation, double* start, int* hits,
                       double* margin_q, double* margin_t, double t_cutoff, double* margin_c) {
  if (n <= 0 || depth < 1 || depth > 64 || n_pix < 0) return -1;
  double* M = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  if (!M) return -2;
#pragma omp parallel for schedule(static)
  for (int g = 0; g < n; ++g) {
    double q[4] = {rot[4 * g], rot[4 * g + 1], rot[4 * g + 2], rot[4 * g + 3]};
    double s[3] = {scale[3 * g], scale[3 * g + 1], scale[3 * g + 2]};
    cov_inverse(q, s, M + 9 * (size_t)g);
  }
  const double o[3] = {cam_pos[0], cam_pos[1], cam_pos[2]};
  const double cq[4] = {cam_rot[0], cam_rot[1], cam_rot[2], cam_rot[3]};
  /* depth + 1 entries decide the image and its margins; up to 8 more let the walk below pass
   * exact t1 ties, as the reference's peel loop does (it peels on until `depth` distinct hits) */
  const int K = depth + 9 > 65 ? 65 : depth + 9;
#pragma omp parallel for schedule(dynamic, 16)
  for (int p = 0; p < n_pix; ++p) {
    /* camera.py:57-71 */
    const double u = ((double)pixels[2 * p] + 0.5) / (double)W, v = ((double)pixels[2 * p + 1] + 0.5) / (double)H;
    double dc[3] = {((double)W * u - 0.5 * (double)W) / (double)fx, ((double)H * v - 0.5 * (double)H) / (double)fy, -1.0};
    const double len = sqrt(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]);
    dc[0] /= len; dc[1] /= len; dc[2] /= len;
    double d[3];
    rot_vec3(cq, dc, d);
    double kt[65], kt2[65];
    int ki[65];
    for (int k = 0; k < K; ++k) { kt[k] = INFINITY; kt2[k] = INFINITY; ki[k] = -1; }
    double mq = INFINITY;
    for (int g = 0; g < n; ++g) {
      const double* Mg = M + 9 * (size_t)g;
      const double m[3] = {o[0] - pos[3 * g], o[1] - pos[3 * g + 1], o[2] - pos[3 * g + 2]};
      const double A = quadform(Mg, d, d);
      const double B = 2.0 * quadform(Mg, d, m);
      const double C = quadform(Mg, m, m) - 3.0;
      const double delta = B * B - 4.0 * A * C;
      if (-B / (2.0 * A) > 0.0) { double w = fabs(delta / (4.0 * A)); if (w < mq) mq = w; }
      if (!(delta > 0.0)) continue; /* delta == 0: t2 = inf, alpha = 0 -- measure zero, skipped */
      const double sq = sqrt(delta);
      const double t1 = (-B - sq) / (2.0 * A), t2 = (-B + sq) / (2.0 * A);
      if (!(t1 > 0.0) || !(t1 < kt[K - 1])) continue; /* scene.py:433: start(0) < t1 < end(inf) */
      int k = K - 1;
      while (k > 0 && kt[k - 1] > t1) { kt[k] = kt[k - 1]; kt2[k] = kt2[k - 1]; ki[k] = ki[k - 1]; --k; }
      kt[k] = t1; kt2[k] = t2; ki[k] = g;
    }
    double L[3] = {0, 0, 0}, T = 1.0, last = 0.0, prev = -1.0, mt = INFINITY, mc = INFINITY;
    int nh = 0;
    double Y[SH_COEFFS];
    if (sh) {
      const double l2 = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      const double dn[3] = {d[0] / l2, d[1] / l2, d[2] / l2}; /* gaussian.py:200 */
      sh_basis(dn, Y);
    }
    for (int k = 0; k < depth; ++k) { /* gap relative to max(1, t1): fp32 error of t1 grows with t1 */
      const double gap = (kt[k + 1] - kt[k]) / fmax(1.0, kt[k + 1]);
      if (isfinite(kt[k + 1]) && gap < mt) mt = gap;
    }
    for (int k = 0; k < K && nh < depth; ++k) {
      if (!isfinite(kt[k])) break;
      if (kt[k] == prev) continue;
      prev = kt[k];
      if (t_cutoff > 0.0) {
        const double c = fabs(T - t_cutoff) / t_cutoff;
        if (c < mc) mc = c;
        if (T < t_cutoff) break; /* terminated: as after a miss */
      }
      const int g = ki[k];
      const double tm = 0.5 * (kt[k] + kt2[k]); /* ray_tracer.py:89 */
      const double x[3] = {o[0] + tm * d[0] - pos[3 * g], o[1] + tm * d[1] - pos[3 * g + 1],
                           o[2] + tm * d[2] - pos[3 * g + 2]};
      const double alpha = (double)opacity[g] * exp(-quadform(M + 9 * (size_t)g, x, x));
      double c[3] = {color[3 * g], color[3 * g + 1], color[3 * g + 2]};
      if (sh)
        for (int b = 0; b < SH_COEFFS; ++b)
          for (int ch = 0; ch < 3; ++ch) c[ch] += Y[b] * (double)sh[((size_t)g * SH_COEFFS + b) * 3 + ch];
      for (int ch = 0; ch < 3; ++ch) L[ch] += T * alpha * c[ch]; /* ray_tracer.py:95-96 */
      T *= 1.0 - alpha;
      last = kt[k];
      ++nh;
    }
    sample[3 * p] = L[0]; sample[3 * p + 1] = L[1]; sample[3 * p + 2] = L[2];
    attenuation[p] = T;
    start[p] = (nh == depth) ? last : INFINITY;
    hits[p] = nh;
    margin_q[p] = mq;
    margin_t[p] = mt;
    if (margin_c) margin_c[p] = mc;
  }
  free(M);
  return 0;
}
