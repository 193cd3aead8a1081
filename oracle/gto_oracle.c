/*
 * gto_oracle.c — CPU restatement (plain C, FP64) of the GTO inner-solve path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (grasptrajopt_amd/, libgto_hip.so) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * Parity status
 *   - building blocks (FK, visual transforms, voxel offsets, SDF value / Jacobian / Hessian,
 *     cost map, seed interpolation) are PINNED against golden vectors produced by executing the
 *     reference's own Python numerics in this container (tests/golden/make_golden.py);
 *   - the ASSEMBLED OBJECTIVES (trajectory: goal-set + standoff, obstacle, velocity terms and the arg-min goal;
 *     IK; base placement), the seed construction / scoring / selection and the parameter marshalling are PINNED
 *     (<= 1e-12 relative) against tests/golden/objective.npz, produced by executing the reference's own
 *     setup_optimization / plan / plan_goalset code (gto/gto_planner.py:42-245, gto/ik_solver.py:30-76,
 *     gto/base_planner.py:35-93) with numeric stand-ins for the CasADi layer
 *     (tests/golden/make_objective_golden.py);
 *   - "parity unpinned": the solver ITERATION only.  IPOPT cannot run here and the reference stores no
 *     input for any of its 853 plans; the projected Levenberg-Marquardt iteration below is
 *     checked against finite differences, the structural invariants of the stored plans and solution-quality
 *     gates, and defines the algorithm the HIP path must reproduce.
 *
 * Every function cites the reference file:line it follows (paths under /root/reference).
 */
#include "../include/gto_solver.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NMAX GTO_MAX_OPT

/* ------------------------------------------------------------------ small matrix helpers */
/* 3x4 affine [R|t] stored row-major in 12 doubles: m[4*r+c]. */
static void aff_identity(double* m) {
  for (int i = 0; i < 12; ++i) m[i] = 0.0;
  m[0] = m[5] = m[10] = 1.0;
}
/* c = a * b  (4x4 homogeneous product restricted to the top 3 rows; optas uses `@`) */
static void aff_mul(const double* a, const double* b, double* c) {
  double r[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j) {
      double s = a[4 * i + 0] * b[0 + j] + a[4 * i + 1] * b[4 + j] + a[4 * i + 2] * b[8 + j];
      if (j == 3) s += a[4 * i + 3];
      r[4 * i + j] = s;
    }
  }
  memcpy(c, r, sizeof r);
}
static void mat3_mul(const double* a, const double* b, double* c) {
  double r[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(c, r, sizeof r);
}

/* optas/spatialmath.py:116-158 rotx/roty/rotz */
static void rotx(double th, double* R) {
  double c = cos(th), s = sin(th);
  double m[9] = {1, 0, 0, 0, c, -s, 0, s, c};
  memcpy(R, m, sizeof m);
}
static void roty(double th, double* R) {
  double c = cos(th), s = sin(th);
  double m[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
  memcpy(R, m, sizeof m);
}
static void rotz(double th, double* R) {
  double c = cos(th), s = sin(th);
  double m[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  memcpy(R, m, sizeof m);
}

/* optas/spatialmath.py:186-211 rpy2r, default order 'zyx': rotz(y) @ roty(p) @ rotx(r) */
void orc_rpy2r(const double rpy[3], double R[9]) {
  double Rz[9], Ry[9], Rx[9], t[9];
  rotz(rpy[2], Rz);
  roty(rpy[1], Ry);
  rotx(rpy[0], Rx);
  mat3_mul(Rz, Ry, t);
  mat3_mul(t, Rx, R);
}

/* optas/spatialmath.py:90-100 angvec2r (Rodrigues) with unit() :293-300 and skew() :228-258 */
void orc_angvec2r(double theta, const double axis[3], double R[9]) {
  double nrm = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  double v0 = axis[0] / nrm, v1 = axis[1] / nrm, v2 = axis[2] / nrm;
  double sk[9] = {0, -v2, v1, v2, 0, -v0, -v1, v0, 0};
  double sk2[9];
  mat3_mul(sk, sk, sk2);
  double s = sin(theta), c1 = 1.0 - cos(theta);
  for (int i = 0; i < 9; ++i) R[i] = s * sk[i] + c1 * sk2[i];
  R[0] += 1.0;
  R[4] += 1.0;
  R[8] += 1.0;
}

/* optas/spatialmath.py:214-225 rt2tr */
static void rt2aff(const double* R, const double* t, double* m) {
  for (int i = 0; i < 3; ++i) {
    m[4 * i + 0] = R[3 * i + 0];
    m[4 * i + 1] = R[3 * i + 1];
    m[4 * i + 2] = R[3 * i + 2];
    m[4 * i + 3] = t[i];
  }
}

/*
 * optas/models.py:826-868 get_global_link_transform, for every frame at once with prefix
 * sharing: T_i = T_parent @ rt2tr(rpy2r(rpy), xyz) [@ joint motion].
 * out: [n_frames][12] affine.
 */
void orc_fk_affine(const gto_robot_desc* d, const double* q, double* out) {
  for (int i = 0; i < d->n_frames; ++i) {
    double O[12], R[9], T[12];
    const double* P;
    double ident[12];
    if (d->parent[i] < 0) {
      aff_identity(ident);
      P = ident;
    } else {
      P = out + 12 * d->parent[i];
    }
    orc_rpy2r(d->origin_rpy + 3 * i, R);
    rt2aff(R, d->origin_xyz + 3 * i, O);
    aff_mul(P, O, T);
    if (d->joint_type[i] == GTO_JOINT_REVOLUTE) {
      double M[12], zero[3] = {0, 0, 0};
      orc_angvec2r(q[d->q_index[i]], d->axis + 3 * i, R); /* :859-860 */
      rt2aff(R, zero, M);
      aff_mul(T, M, T);
    } else if (d->joint_type[i] == GTO_JOINT_PRISMATIC) {
      const double* ax = d->axis + 3 * i;
      double nrm = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      double qi = q[d->q_index[i]];
      double tr[3] = {qi * (ax[0] / nrm), qi * (ax[1] / nrm), qi * (ax[2] / nrm)}; /* :862-863 */
      double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, M[12];
      rt2aff(I, tr, M);
      aff_mul(T, M, T);
    }
    memcpy(out + 12 * i, T, sizeof T);
  }
}

static void aff_to_44(const double* a, double* m) {
  memcpy(m, a, 12 * sizeof(double));
  m[12] = m[13] = m[14] = 0.0;
  m[15] = 1.0;
}

/* frames_out [nq][n_frames][16] */
void orc_eval_fk(const gto_robot_desc* d, int nq, const double* q, double* frames_out) {
  double* a = (double*)malloc(sizeof(double) * 12 * d->n_frames);
  for (int k = 0; k < nq; ++k) {
    orc_fk_affine(d, q + (size_t)k * d->ndof, a);
    for (int i = 0; i < d->n_frames; ++i)
      aff_to_44(a + 12 * i, frames_out + ((size_t)k * d->n_frames + i) * 16);
  }
  free(a);
}

/* gto/gto_models.py:92-100: visual_tf = link_tf @ rt2tr(rpy2r(vis_rpy), vis_xyz); out [n_links][12] */
void orc_visual_affine(const gto_robot_desc* d, const double* frames12, double* out) {
  for (int l = 0; l < d->n_links; ++l) {
    double R[9], V[12];
    orc_rpy2r(d->visual_rpy + 3 * l, R);
    rt2aff(R, d->visual_xyz + 3 * l, V);
    aff_mul(frames12 + 12 * d->link_frame[l], V, out + 12 * l);
  }
}

/* ------------------------------------------------------------------ voxel field */
typedef struct orc_field {
  const float* c_all;
  const float* c_obs;
  int32_t shape[3];
  double origin[3];
  double res;
} orc_field;

/* gto/gto_models.py:174-187 points_to_offsets: floor((x-origin)/res), clip to [0,N-1],
 * off = iz + Nz*(iy + Ny*ix).  (The numpy twin :190-201 truncates after clipping: same result.) */
static inline void voxel_index(const orc_field* f, const double x[3], int32_t idx[3]) {
  for (int a = 0; a < 3; ++a) {
    double v = floor((x[a] - f->origin[a]) / f->res);
    double hi = (double)(f->shape[a] - 1);
    if (!(v >= 0.0)) v = 0.0; /* fmax(v,0); NaN -> 0 */
    if (v > hi) v = hi;
    idx[a] = (int32_t)v;
  }
}
static inline int64_t flat_offset(const orc_field* f, int32_t ix, int32_t iy, int32_t iz) {
  return (int64_t)iz + (int64_t)f->shape[2] * ((int64_t)iy + (int64_t)f->shape[1] * (int64_t)ix);
}
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_points_to_offsets(const double* xyz, int n, const double origin[3], double res,
                           const int32_t shape[3], int32_t* off) {
  orc_field f;
  memset(&f, 0, sizeof f);
  memcpy(f.shape, shape, sizeof f.shape);
  memcpy(f.origin, origin, sizeof f.origin);
  f.res = res;
  for (int i = 0; i < n; ++i) {
    int32_t idx[3];
    voxel_index(&f, xyz + 3 * i, idx);
    off[i] = (int32_t)flat_offset(&f, idx[0], idx[1], idx[2]);
  }
}

/* gto/sdf_callback.py:43-49 SDFCallback.eval */
static inline double field_value(const orc_field* f, const float* data, const int32_t idx[3]) {
  return (double)data[flat_offset(f, idx[0], idx[1], idx[2])];
}
/* gto/sdf_callback.py:90-114 JacFun.eval: neighbours clipped, divisor stays 2*res.
 * (The division is done as a multiplication by 1/(2 res); values agree to 1 ulp.) */
static inline void field_grad(const orc_field* f, const float* data, const int32_t idx[3], double g[3]) {
  double inv2r = 1.0 / (2.0 * f->res);
  for (int a = 0; a < 3; ++a) {
    int32_t p[3] = {idx[0], idx[1], idx[2]}, m[3] = {idx[0], idx[1], idx[2]};
    p[a] = clampi(idx[a] + 1, 0, f->shape[a] - 1);
    m[a] = clampi(idx[a] - 1, 0, f->shape[a] - 1);
    g[a] = ((double)data[flat_offset(f, p[0], p[1], p[2])] - (double)data[flat_offset(f, m[0], m[1], m[2])]) * inv2r;
  }
}

/* value / Jacobian / Hessian of a field at arbitrary points, exactly as the three sdf_callback
 * classes compute them (gto/sdf_callback.py:43-49, 90-114, 159-183). Any output may be NULL. */
void orc_sdf_eval(const float* data, const int32_t shape[3], const double origin[3], double res, int n,
                  const double* xyz, double* val, double* jac, double* hes) {
  orc_field f;
  memset(&f, 0, sizeof f);
  memcpy(f.shape, shape, sizeof f.shape);
  memcpy(f.origin, origin, sizeof f.origin);
  f.res = res;
  for (int i = 0; i < n; ++i) {
    int32_t idx[3];
    voxel_index(&f, xyz + 3 * i, idx);
    if (val) val[i] = field_value(&f, data, idx);
    if (jac) {
      /* exact division here: this entry point is compared with the reference bit-for-bit-ish */
      for (int a = 0; a < 3; ++a) {
        int32_t p[3] = {idx[0], idx[1], idx[2]}, m[3] = {idx[0], idx[1], idx[2]};
        p[a] = clampi(idx[a] + 1, 0, shape[a] - 1);
        m[a] = clampi(idx[a] - 1, 0, shape[a] - 1);
        jac[3 * i + a] = ((double)data[flat_offset(&f, p[0], p[1], p[2])] -
                          (double)data[flat_offset(&f, m[0], m[1], m[2])]) / (2.0 * res);
      }
    }
    if (hes) {
      for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) {
          double v[4];
          int k = 0;
          for (int sa = 1; sa >= -1; sa -= 2)
            for (int sb = 1; sb >= -1; sb -= 2) {
              int32_t j[3] = {idx[0], idx[1], idx[2]};
              j[a] += sa;
              j[b] += sb;
              for (int c = 0; c < 3; ++c) j[c] = clampi(j[c], 0, shape[c] - 1); /* get_value :159-162 */
              v[k++] = (double)data[flat_offset(&f, j[0], j[1], j[2])];
            }
          /* f1 = (+,+), f2 = (+,-), f3 = (-,+), f4 = (-,-); (f1-f2-f3+f4)/(4 res^2) :176-180 */
          double h = (v[0] - v[1] - v[2] + v[3]) / (4.0 * res * res);
          hes[9 * i + 3 * a + b] = h;
          hes[9 * i + 3 * b + a] = h;
        }
    }
  }
}

/* mesh_to_sdf/depth_point_cloud.py:84-89 cost map, float32 arithmetic like the numpy arrays there:
 * inside -> w_inside*(-d + eps/2); 0<d<eps -> (d-eps)^2/(2 eps); else 0.  d is the signed distance
 * (already negated for inside points, :70-71). */
void orc_sdf_cost_map(int n, const float* signed_dist, const unsigned char* inside, float epsilon,
                      float w_inside, float* cost) {
  for (int i = 0; i < n; ++i) {
    float d = signed_dist[i];
    float c = 0.0f;
    if (inside[i]) {
      c = w_inside * (-d + epsilon / 2.0f);
    } else if (d > 0.0f && d < epsilon) {
      float e = d - epsilon;
      c = (e * e) / (2.0f * epsilon);
    }
    cost[i] = c;
  }
}

/* ------------------------------------------------------------------ cost field from a depth image
 * mesh_to_sdf/depth_point_cloud.py:9-141 DepthPointCloud: the producer of the (F,) cost arrays.
 *   backproject (:32-52): X = depth * (Kinv @ [x, y, 1]) for pixels with 0 < depth < threshold (and
 *   target_mask == 0), row-major pixel order; points = R_cam X + t_cam (:21-23).
 *   get_sdf (:56-61): distance to the nearest cloud point (sklearn KDTree, float64 -> float32), negated
 *   where the query is not "outside".
 *   is_outside (:126-141): project the query into the camera; inside the viewport it is outside iff its
 *   camera depth is smaller than the depth image at the pixel (int() truncation); outside the viewport: outside.
 * Kinv and cam_inv are passed in (the reference gets them from np.linalg.inv; a caller that wants bit
 * parity passes the same). Arithmetic follows the reference's order, without FMA contraction (-ffp-contract=off). */
/* points_out [H*W][3] in pixel order with valid_out flags; returns the number of valid points */
int orc_depth_backproject(const float* depth, int H, int W, const double* Kinv, const double* cam,
                          const unsigned char* target_mask, double threshold, double* points_out,
                          unsigned char* valid_out) {
  int n = 0;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const int i = y * W + x;
      const float d = depth[i];
      const int ok = (d > 0.0f) && ((double)d < threshold) && (!target_mask || target_mask[i] == 0);
      valid_out[i] = (unsigned char)ok;
      /* R = Kinv @ [x, y, 1] (float64), X = depth (f32 -> f64) * R */
      double X[3];
      for (int r = 0; r < 3; ++r) X[r] = (double)d * (Kinv[3 * r] * (double)x + Kinv[3 * r + 1] * (double)y + Kinv[3 * r + 2] * 1.0);
      for (int r = 0; r < 3; ++r)
        points_out[3 * (size_t)i + r] = (cam[4 * r] * X[0] + cam[4 * r + 1] * X[1] + cam[4 * r + 2] * X[2]) + cam[4 * r + 3];
      n += ok;
    }
  return n;
}

/* signed distance (float32) and inside flag of nq query points against the valid cloud points (brute force) */
void orc_depth_sdf(const double* points, const unsigned char* valid, int H, int W, const float* depth, const double* K,
                   const double* cam_inv, const double* query, long nq, float* sdf_out, unsigned char* inside_out) {
  const long N = (long)H * W;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (long q = 0; q < nq; ++q) {
    const double* p = query + 3 * q;
    double best = INFINITY;
    for (long i = 0; i < N; ++i) {
      if (!valid[i]) continue;
      const double dx = p[0] - points[3 * i], dy = p[1] - points[3 * i + 1], dz = p[2] - points[3 * i + 2];
      const double d2 = (dx * dx + dy * dy) + dz * dz;
      if (d2 < best) best = d2;
    }
    float dist = (float)sqrt(best);
    /* is_outside */
    double pc[3];
    for (int r = 0; r < 3; ++r) pc[r] = (cam_inv[4 * r] * p[0] + cam_inv[4 * r + 1] * p[1] + cam_inv[4 * r + 2] * p[2]) + cam_inv[4 * r + 3];
    double u[3];
    for (int r = 0; r < 3; ++r) u[r] = K[3 * r] * pc[0] + K[3 * r + 1] * pc[1] + K[3 * r + 2] * pc[2];
    const double ux = u[0] / u[2], uy = u[1] / u[2];
    /* .astype(int): truncation toward zero; non-finite or out-of-range values become INT64_MIN in NumPy */
    long px = (ux == ux && fabs(ux) < 9.0e18) ? (long)ux : LONG_MIN;
    long py = (uy == uy && fabs(uy) < 9.0e18) ? (long)uy : LONG_MIN;
    int outside = 1;
    if (px >= 0 && py >= 0 && px < W && py < H) outside = pc[2] < (double)depth[py * W + px];
    if (!outside) dist = -dist;
    if (sdf_out) sdf_out[q] = dist;
    if (inside_out) inside_out[q] = (unsigned char)!outside;
  }
}

/* gto/utils.py:63-82 interpolate_waypoints for the two-waypoint call the planner makes
 * (gto/gto_planner.py:155,203): scipy CubicSpline(bc_type="clamped") through (0,y0),(1,y1) is the
 * Hermite cubic with zero end slopes, sampled at linspace(0,1,n+2)[1:-1] (endpoints excluded).
 * waypoints [2][m] -> out [n][m]. */
void orc_interpolate_waypoints(const double* waypoints, int n, int m, double* out) {
  for (int k = 0; k < n; ++k) {
    double s = (double)(k + 1) / (double)(n + 1);
    double h = s * s * (3.0 - 2.0 * s);
    for (int j = 0; j < m; ++j) {
      double y0 = waypoints[j], y1 = waypoints[m + j];
      out[(size_t)k * m + j] = y0 + (y1 - y0) * h;
    }
  }
}

/* ------------------------------------------------------------------ problem evaluation */
typedef struct orc_instance {
  const gto_robot_desc* d;
  const gto_solver_opts* o;
  orc_field field;
  int n_goals;
  const double* goals;    /* [n_goals][16] */
  const double* standoff; /* [16] or NULL */
  double base[3];
  const double* Qfull; /* [ndof][T] provides the parameter-joint rows */
} orc_instance;

/* per-evaluation results */
typedef struct orc_eval {
  double f_goal, f_obs, f_vel; /* weighted terms of Appendix A */
  int goal_argmin;
  double* JtJ_obs;  /* [T][n][n] unweighted */
  double* Jtr_obs;  /* [T][n] */
  double* sumsq;    /* [T] */
  double JtJ_goal[2][NMAX * NMAX]; /* 0: final waypoint, 1: standoff waypoint */
  double Jtr_goal[2][NMAX];
} orc_eval;

static void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* which optimised joints move frame f, plus their world axis / origin at this configuration */
typedef struct orc_kin {
  double frames[GTO_MAX_FRAMES * 12];
  double vis[GTO_MAX_LINKS * 12];
  double axis_w[NMAX][3]; /* world axis of optimised joint j */
  double org_w[NMAX][3];  /* a point on that axis             */
  int jtype[NMAX];
  unsigned anc[GTO_MAX_FRAMES]; /* bit j set: optimised joint j is an ancestor-or-self of the frame */
} orc_kin;

static void kin_compute(const gto_robot_desc* d, const double* q, orc_kin* k) {
  orc_fk_affine(d, q, k->frames);
  orc_visual_affine(d, k->frames, k->vis);
  for (int i = 0; i < d->n_frames; ++i) {
    unsigned m = d->parent[i] >= 0 ? k->anc[d->parent[i]] : 0u;
    if (d->q_index[i] >= 0) {
      for (int j = 0; j < d->n_opt; ++j)
        if (d->opt_index[j] == d->q_index[i]) {
          m |= 1u << j;
          const double* T = k->frames + 12 * i;
          const double* ax = d->axis + 3 * i;
          double nrm = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
          double u[3] = {ax[0] / nrm, ax[1] / nrm, ax[2] / nrm};
          for (int r = 0; r < 3; ++r) {
            k->axis_w[j][r] = T[4 * r] * u[0] + T[4 * r + 1] * u[1] + T[4 * r + 2] * u[2];
            k->org_w[j][r] = T[4 * r + 3];
          }
          k->jtype[j] = d->joint_type[i];
        }
    }
    k->anc[i] = m;
  }
}

/* d(point y rigidly attached to a frame with ancestor mask `anc`)/d q_j, robot-base coordinates */
static inline void point_jac_col(const orc_kin* k, int j, const double* y, double* col) {
  if (k->jtype[j] == GTO_JOINT_PRISMATIC) {
    col[0] = k->axis_w[j][0];
    col[1] = k->axis_w[j][1];
    col[2] = k->axis_w[j][2];
  } else {
    double r[3] = {y[0] - k->org_w[j][0], y[1] - k->org_w[j][1], y[2] - k->org_w[j][2]};
    cross3(k->axis_w[j], r, col);
  }
}

static void full_q(const orc_instance* in, const double* Qopt, int t, double* q) {
  const gto_robot_desc* d = in->d;
  int T = in->o->T;
  for (int i = 0; i < d->ndof; ++i) q[i] = in->Qfull[(size_t)i * T + t];
  for (int j = 0; j < d->n_opt; ++j) q[d->opt_index[j]] = Qopt[(size_t)j * T + t];
}

/* target pose of the gripper link for goal RT: RT @ [S @] G, G = invt(T_ee) @ T_gripper
 * (gto/gto_planner.py:93-101, gto/gto_planner.py:38, optas/models.py:900-902) */
static void goal_target(const orc_kin* k, const gto_robot_desc* d, const double* RT16,
                        const double* S16, double* Y) {
  const double* Te = k->frames + 12 * d->frame_ee;
  const double* Tg = k->frames + 12 * d->frame_gripper;
  double inv[12], G[12];
  for (int r = 0; r < 3; ++r) { /* invt: [R^T | -R^T t] (optas/spatialmath.py:271-280) */
    for (int c = 0; c < 3; ++c) inv[4 * r + c] = Te[4 * c + r];
    inv[4 * r + 3] = -(Te[0 + r] * Te[3] + Te[4 + r] * Te[7] + Te[8 + r] * Te[11]);
  }
  aff_mul(inv, Tg, G);
  double RT[12];
  memcpy(RT, RT16, sizeof RT);
  if (S16) {
    double S[12];
    memcpy(S, S16, sizeof S);
    aff_mul(RT, S, RT);
  }
  aff_mul(RT, G, Y);
}

static inline void aff_apply(const double* A, const double* p, double* x) {
  for (int r = 0; r < 3; ++r) x[r] = A[4 * r] * p[0] + A[4 * r + 1] * p[1] + A[4 * r + 2] * p[2] + A[4 * r + 3];
}

/* Evaluate everything at trajectory Qopt [n][T].  want_deriv: also the Gauss-Newton blocks. */
static void evaluate(const orc_instance* in, const double* Qopt, int want_deriv, orc_eval* ev) {
  const gto_robot_desc* d = in->d;
  const gto_solver_opts* o = in->o;
  const int T = o->T, n = d->n_opt;
  const int ts = T + o->standoff_offset;
  const double dt = o->Tmax / (double)(T - 1);
  double q[GTO_MAX_DOF];
  orc_kin* k = (orc_kin*)malloc(sizeof(orc_kin));
  double fobs = 0.0;

  if (want_deriv) {
    memset(ev->JtJ_obs, 0, sizeof(double) * (size_t)T * n * n);
    memset(ev->Jtr_obs, 0, sizeof(double) * (size_t)T * n);
    memset(ev->JtJ_goal, 0, sizeof ev->JtJ_goal);
    memset(ev->Jtr_goal, 0, sizeof ev->Jtr_goal);
  }
  double* goal_cost = (double*)calloc((size_t)in->n_goals, sizeof(double));

  for (int t = 0; t < T; ++t) {
    full_q(in, Qopt, t, q);
    kin_compute(d, q, k);
    /* obstacle term (gto/gto_planner.py:108-131): c_all before the standoff waypoint, c_obs after */
    const float* data = (t < ts) ? in->field.c_all : in->field.c_obs;
    double ss = 0.0;
    for (int p = 0; p < d->n_points; ++p) {
      int l = d->point_link[p];
      double y[3], x[3];
      aff_apply(k->vis + 12 * l, d->points + 3 * p, y);
      x[0] = y[0] + in->base[0];
      x[1] = y[1] + in->base[1];
      x[2] = y[2] + in->base[2];
      int32_t idx[3];
      voxel_index(&in->field, x, idx);
      double c = field_value(&in->field, data, idx);
      ss += c * c;
      if (want_deriv && o->grad_mode == GTO_GRAD_CENTRAL_DIFF) {
        double g[3];
        field_grad(&in->field, data, idx, g);
        if (g[0] != 0.0 || g[1] != 0.0 || g[2] != 0.0) {
          unsigned anc = k->anc[d->link_frame[l]];
          double J[NMAX];
          for (int j = 0; j < n; ++j) {
            J[j] = 0.0;
            if (anc >> j & 1u) {
              double col[3];
              point_jac_col(k, j, y, col);
              J[j] = g[0] * col[0] + g[1] * col[1] + g[2] * col[2];
            }
          }
          double* A = ev->JtJ_obs + (size_t)t * n * n;
          double* b = ev->Jtr_obs + (size_t)t * n;
          for (int i = 0; i < n; ++i) {
            b[i] += J[i] * c;
            for (int j = 0; j < n; ++j) A[i * n + j] += J[i] * J[j];
          }
        }
      }
    }
    ev->sumsq[t] = ss;
    fobs += ss;

    /* goal-set point matching at the last waypoint and (optionally) the standoff waypoint
     * (gto/gto_planner.py:84-105) */
    for (int which = 0; which < 2; ++which) {
      if (which == 0 && t != T - 1) continue;
      if (which == 1 && !(in->standoff && t == ts)) continue;
      const double* A = k->frames + 12 * d->frame_gripper;
      for (int g = 0; g < in->n_goals; ++g) {
        double Y[12];
        goal_target(k, d, in->goals + 16 * g, which == 1 ? in->standoff : NULL, Y);
        double s = 0.0;
        for (int p = 0; p < d->n_gripper_points; ++p) {
          double xa[3], xb[3];
          aff_apply(A, d->gripper_points + 3 * p, xa);
          aff_apply(Y, d->gripper_points + 3 * p, xb);
          for (int r = 0; r < 3; ++r) s += (xa[r] - xb[r]) * (xa[r] - xb[r]);
        }
        goal_cost[g] += s;
      }
    }
  }

  /* optas.mmin over the goal set (gto/gto_planner.py:105): first minimum.  A goal whose cost is not a finite number
   * (NaN / Inf in its pose) never wins against one that is; a set without a finite cost yields goal 0 and f_goal = +Inf,
   * which ends the solve with GTO_STATUS_NUMERICAL at the first evaluation (solve_instance). */
  int best = -1;
  double best_c = INFINITY;
  for (int g = 0; g < in->n_goals; ++g)
    if (goal_cost[g] < best_c) {
      best = g;
      best_c = goal_cost[g];
    }
  if (best < 0) best = 0;
  ev->goal_argmin = best;
  ev->f_goal = best_c;
  ev->f_obs = o->w_obstacle * fobs;

  /* velocity term (gto/gto_planner.py:133-135) with the velocities eliminated through the linear
   * dynamics: dQ_0 = 0, dQ_t = (Q_{t+1}-Q_t)/dt */
  double fv = 0.0;
  for (int t = 1; t < T - 1; ++t)
    for (int j = 0; j < n; ++j) {
      double v = (Qopt[(size_t)j * T + t + 1] - Qopt[(size_t)j * T + t]) / dt;
      fv += v * v;
    }
  ev->f_vel = o->w_vel * fv;

  if (want_deriv) {
    /* Gauss-Newton blocks of the arg-min goal */
    for (int which = 0; which < 2; ++which) {
      if (which == 1 && !in->standoff) break;
      int t = which == 0 ? T - 1 : ts;
      full_q(in, Qopt, t, q);
      kin_compute(d, q, k);
      const double* A = k->frames + 12 * d->frame_gripper;
      unsigned anc = k->anc[d->frame_gripper];
      double Y[12];
      goal_target(k, d, in->goals + 16 * best, which == 1 ? in->standoff : NULL, Y);
      double* H = ev->JtJ_goal[which];
      double* gvec = ev->Jtr_goal[which];
      for (int p = 0; p < d->n_gripper_points; ++p) {
        double xa[3], xb[3], r[3], J[3][NMAX];
        aff_apply(A, d->gripper_points + 3 * p, xa);
        aff_apply(Y, d->gripper_points + 3 * p, xb);
        for (int c = 0; c < 3; ++c) r[c] = xa[c] - xb[c];
        for (int j = 0; j < n; ++j) {
          double col[3] = {0, 0, 0};
          if (anc >> j & 1u) point_jac_col(k, j, xa, col);
          J[0][j] = col[0];
          J[1][j] = col[1];
          J[2][j] = col[2];
        }
        for (int i = 0; i < n; ++i) {
          gvec[i] += J[0][i] * r[0] + J[1][i] * r[1] + J[2][i] * r[2];
          for (int j = 0; j < n; ++j)
            H[i * n + j] += J[0][i] * J[0][j] + J[1][i] * J[1][j] + J[2][i] * J[2][j];
        }
      }
    }
  }
  free(goal_cost);
  free(k);
}

/* ------------------------------------------------------------------ block-tridiagonal solve */
/* In-place Cholesky of an n x n SPD block (lower); returns 0 on a non-positive pivot. */
static int chol(double* S, int n) {
  for (int j = 0; j < n; ++j) {
    double d = S[j * n + j];
    for (int k = 0; k < j; ++k) d -= S[j * n + k] * S[j * n + k];
    if (!(d > 0.0)) return 0;
    d = sqrt(d);
    S[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = S[i * n + j];
      for (int k = 0; k < j; ++k) s -= S[i * n + k] * S[j * n + k];
      S[i * n + j] = s / d;
    }
  }
  return 1;
}

/*
 * Solve  M x = rhs,  M block tridiagonal over waypoints t = 0..m-1 with dense diagonal blocks
 * D[t] (n x n) and DIAGONAL coupling blocks diag(e[t]) between t and t+1.
 * L [m][n][n] and C [m][n][n] are workspaces. Returns 0 if M is not positive definite.
 */
static int block_tridiag_solve(int m, int n, const double* D, const double* e, const double* rhs,
                               double* x, double* L, double* C) {
  for (int t = 0; t < m; ++t) {
    double* Lt = L + (size_t)t * n * n;
    double* Ct = C + (size_t)t * n * n;
    memcpy(Lt, D + (size_t)t * n * n, sizeof(double) * n * n);
    if (t > 0) {
      const double* Lp = L + (size_t)(t - 1) * n * n;
      const double* ep = e + (size_t)(t - 1) * n;
      /* C_t = E_{t-1} L_{t-1}^{-T}: row i of C solves L_{t-1} c = e_i (unit vector scaled by ep[i]) */
      for (int i = 0; i < n; ++i) {
        for (int c = 0; c < n; ++c) {
          double s = (c == i) ? ep[i] : 0.0;
          for (int k = 0; k < c; ++k) s -= Lp[c * n + k] * Ct[i * n + k];
          Ct[i * n + c] = s / Lp[c * n + c];
        }
      }
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = 0.0;
          for (int k = 0; k < n; ++k) s += Ct[i * n + k] * Ct[j * n + k];
          Lt[i * n + j] -= s;
        }
    }
    if (!chol(Lt, n)) return 0;
    /* forward substitution y_t = L_t^{-1} (rhs_t - C_t y_{t-1}) */
    double* yt = x + (size_t)t * n;
    for (int i = 0; i < n; ++i) {
      double s = rhs[(size_t)t * n + i];
      if (t > 0)
        for (int k = 0; k < n; ++k) s -= Ct[i * n + k] * x[(size_t)(t - 1) * n + k];
      yt[i] = s;
    }
    for (int i = 0; i < n; ++i) {
      double s = yt[i];
      for (int k = 0; k < i; ++k) s -= Lt[i * n + k] * yt[k];
      yt[i] = s / Lt[i * n + i];
    }
  }
  for (int t = m - 1; t >= 0; --t) {
    const double* Lt = L + (size_t)t * n * n;
    double* xt = x + (size_t)t * n;
    if (t < m - 1) {
      const double* Cn = C + (size_t)(t + 1) * n * n;
      for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += Cn[k * n + i] * x[(size_t)(t + 1) * n + k];
        xt[i] -= s;
      }
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = xt[i];
      for (int k = i + 1; k < n; ++k) s -= Lt[k * n + i] * xt[k];
      xt[i] = s / Lt[i * n + i];
    }
  }
  return 1;
}

/* ------------------------------------------------------------------ the solver */
typedef struct orc_work {
  int T, n;
  orc_eval cur, tri;
  double *Q, *Qtry;          /* [n][T] */
  double *D, *e, *rhs, *dx;  /* free waypoints m = T-2 */
  double *L, *C, *Aun;       /* workspaces; Aun = undamped diagonal blocks */
  double* bfull;             /* undamped gradient half b = J^T r */
  int* act;                  /* active-set flags */
} orc_work;

static void eval_alloc(orc_eval* ev, int T, int n) {
  ev->JtJ_obs = (double*)calloc((size_t)T * n * n, sizeof(double));
  ev->Jtr_obs = (double*)calloc((size_t)T * n, sizeof(double));
  ev->sumsq = (double*)calloc((size_t)T, sizeof(double));
}
static void eval_free(orc_eval* ev) {
  free(ev->JtJ_obs);
  free(ev->Jtr_obs);
  free(ev->sumsq);
}
static void eval_swap(orc_eval* a, orc_eval* b) {
  orc_eval t = *a;
  *a = *b;
  *b = t;
}

/*
 * Bound-constrained Levenberg-Marquardt / Gauss-Newton on the eliminated problem of SURVEY.md
 * Appendix A: free variables Q[:, 2..T-1]; Q[:,0] = Q[:,1] = qc[opt] (initial configuration +
 * zero initial velocity + Euler dynamics, gto/gto_planner.py:59-72); lo <= Q <= hi (:138).
 * Stands in for IPOPT (optas/solver.py:384-400, options gto/gto_planner.py:141).
 * One "iteration" = one evaluation of a trial trajectory + one accept/reject + one step solve.
 */
static void solve_instance(const orc_instance* in, const double* qc, double* Q_out, double* dQ_out,
                           double* cost_out, int32_t* iters_out, int32_t* status_out,
                           double* f_trace /* [max_iter+1] or NULL */) {
  const gto_robot_desc* d = in->d;
  const gto_solver_opts* o = in->o;
  const int T = o->T, n = d->n_opt, m = T - 2;
  const int ts = T + o->standoff_offset;
  const double dt = o->Tmax / (double)(T - 1);
  const double alpha = o->w_vel / (dt * dt);

  orc_work w;
  w.T = T;
  w.n = n;
  eval_alloc(&w.cur, T, n);
  eval_alloc(&w.tri, T, n);
  w.Q = (double*)malloc(sizeof(double) * n * T);
  w.Qtry = (double*)malloc(sizeof(double) * n * T);
  w.D = (double*)malloc(sizeof(double) * m * n * n);
  w.Aun = (double*)malloc(sizeof(double) * m * n * n);
  w.e = (double*)malloc(sizeof(double) * m * n);
  w.rhs = (double*)malloc(sizeof(double) * m * n);
  w.dx = (double*)malloc(sizeof(double) * m * n);
  w.L = (double*)malloc(sizeof(double) * m * n * n);
  w.C = (double*)malloc(sizeof(double) * m * n * n);
  w.bfull = (double*)malloc(sizeof(double) * m * n);
  w.act = (int*)malloc(sizeof(int) * m * n);

  /* seed: optimised rows of Q0, first two waypoints pinned to qc, the rest clipped into bounds */
  for (int j = 0; j < n; ++j) {
    double q0 = qc[d->opt_index[j]];
    for (int t = 0; t < T; ++t) {
      double v = in->Qfull[(size_t)d->opt_index[j] * T + t];
      if (t < 2) v = q0;
      else {
        if (v < d->lower[j]) v = d->lower[j];
        if (v > d->upper[j]) v = d->upper[j];
      }
      w.Qtry[(size_t)j * T + t] = v;
    }
  }

  double lambda = o->lambda0, nu = 2.0, f = INFINITY, pred = 0.0;
  int status = GTO_STATUS_MAX_ITER, first = 1, k = 0;
  for (;; ++k) {
    evaluate(in, w.Qtry, 1, &w.tri);
    double f_try = w.tri.f_goal + w.tri.f_obs + w.tri.f_vel;
    if (f_trace) f_trace[k] = f_try;
    int done = 0;
    if (first) {
      first = 0;
      memcpy(w.Q, w.Qtry, sizeof(double) * n * T);
      f = f_try;
      eval_swap(&w.cur, &w.tri);
      /* a seed whose objective is not a finite number (NaN / Inf in a goal pose, in the seed, in a voxel the seed touches):
       * nothing to descend from -- the iterate is returned as it is (optas/solver.py:135), status NUMERICAL, 0 iterations.
       * (A TRIAL point with such an objective is a rejected step: f_try < f is false.) */
      if (!isfinite(f)) {
        status = GTO_STATUS_NUMERICAL;
        done = 1;
      }
    } else if (f_try < f && pred > 0.0) {
      double df = f - f_try, rho = df / pred;
      memcpy(w.Q, w.Qtry, sizeof(double) * n * T);
      f = f_try;
      eval_swap(&w.cur, &w.tri);
      double s = 2.0 * rho - 1.0, fac = 1.0 - s * s * s;
      if (fac < 1.0 / 3.0) fac = 1.0 / 3.0;
      lambda *= fac;
      if (lambda < 1e-12) lambda = 1e-12;
      nu = 2.0;
      if (df <= o->tol_rel_f * (1.0 + f)) {
        status = GTO_STATUS_CONVERGED;
        done = 1;
      }
    } else {
      lambda *= nu;
      nu *= 2.0;
      if (lambda > 1e15) {
        status = GTO_STATUS_CONVERGED; /* no descent direction left on the stepwise objective */
        done = 1;
      }
    }
    if (done) break;
    if (k >= o->max_iter) {
      status = GTO_STATUS_MAX_ITER;
      break;
    }

    /* normal equations at the current iterate: A = J^T J, b = J^T r (f = sum r^2).
     * Aun / bfull: undamped model over the free waypoints t = 2..T-1 (slot t-2). */
    for (int t = 2; t < T; ++t) {
      double* A = w.Aun + (size_t)(t - 2) * n * n;
      double* b = w.bfull + (size_t)(t - 2) * n;
      const double* Ao = w.cur.JtJ_obs + (size_t)t * n * n;
      const double* bo = w.cur.Jtr_obs + (size_t)t * n;
      for (int i = 0; i < n; ++i) {
        b[i] = o->w_obstacle * bo[i];
        for (int j = 0; j < n; ++j) A[i * n + j] = o->w_obstacle * Ao[i * n + j];
      }
      for (int which = 0; which < 2; ++which) {
        if (which == 0 && t != T - 1) continue;
        if (which == 1 && !(in->standoff && t == ts)) continue;
        for (int i = 0; i < n; ++i) {
          b[i] += w.cur.Jtr_goal[which][i];
          for (int j = 0; j < n; ++j) A[i * n + j] += w.cur.JtJ_goal[which][i * n + j];
        }
      }
      /* velocity residuals sqrt(alpha) (Q_{t+1}-Q_t): pairs (t-1,t) and (t,t+1) */
      for (int i = 0; i < n; ++i) {
        double qt = w.Q[(size_t)i * T + t], qm = w.Q[(size_t)i * T + t - 1];
        A[i * n + i] += alpha;
        b[i] += alpha * (qt - qm);
        if (t < T - 1) {
          double qp = w.Q[(size_t)i * T + t + 1];
          A[i * n + i] += alpha;
          b[i] -= alpha * (qp - qt);
        }
      }
    }
    /* active set: a variable sitting on a bound whose descent direction points outward is frozen */
    for (int t = 2; t < T; ++t)
      for (int i = 0; i < n; ++i) {
        double qv = w.Q[(size_t)i * T + t], bi = w.bfull[(size_t)(t - 2) * n + i];
        w.act[(size_t)(t - 2) * n + i] = (qv <= d->lower[i] && bi > 0.0) || (qv >= d->upper[i] && bi < 0.0);
      }
    /* damped system: D = A with Marquardt scaling (1+lambda) on the diagonal, frozen rows/cols -> identity */
    for (int t = 2; t < T; ++t) {
      const double* A = w.Aun + (size_t)(t - 2) * n * n;
      double* Dm = w.D + (size_t)(t - 2) * n * n;
      const int* act = w.act + (size_t)(t - 2) * n;
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
          double v = A[i * n + j];
          if (act[i] || act[j]) v = (i == j) ? 1.0 : 0.0;
          else if (i == j) v *= (1.0 + lambda);
          Dm[i * n + j] = v;
        }
        w.rhs[(size_t)(t - 2) * n + i] = act[i] ? 0.0 : -w.bfull[(size_t)(t - 2) * n + i];
        /* coupling between waypoint t and t+1 (slot t-2) */
        int a1 = (t < T - 1) ? w.act[(size_t)(t - 1) * n + i] : 1;
        w.e[(size_t)(t - 2) * n + i] = (act[i] || a1) ? 0.0 : -alpha;
      }
    }
    if (!block_tridiag_solve(m, n, w.D, w.e, w.rhs, w.dx, w.L, w.C)) {
      status = GTO_STATUS_NUMERICAL;
      break;
    }
    /* trial point, projected onto the bounds */
    double maxstep = 0.0;
    memcpy(w.Qtry, w.Q, sizeof(double) * n * T);
    for (int t = 2; t < T; ++t)
      for (int i = 0; i < n; ++i) {
        double v = w.Q[(size_t)i * T + t] + w.dx[(size_t)(t - 2) * n + i];
        if (v < d->lower[i]) v = d->lower[i];
        if (v > d->upper[i]) v = d->upper[i];
        w.Qtry[(size_t)i * T + t] = v;
        double s = v - w.Q[(size_t)i * T + t];
        w.dx[(size_t)(t - 2) * n + i] = s;
        if (fabs(s) > maxstep) maxstep = fabs(s);
      }
    if (maxstep < o->tol_step) {
      status = GTO_STATUS_CONVERGED;
      break;
    }
    /* predicted decrease of the undamped Gauss-Newton model: -(2 b^T s + s^T A s) */
    double bts = 0.0, sAs = 0.0;
    for (int t = 2; t < T; ++t) {
      const double* A = w.Aun + (size_t)(t - 2) * n * n;
      const double* s = w.dx + (size_t)(t - 2) * n;
      for (int i = 0; i < n; ++i) {
        bts += w.bfull[(size_t)(t - 2) * n + i] * s[i];
        double r = 0.0;
        for (int j = 0; j < n; ++j) r += A[i * n + j] * s[j];
        sAs += s[i] * r;
        if (t < T - 1) sAs += 2.0 * (-alpha) * s[i] * w.dx[(size_t)(t - 1) * n + i];
      }
    }
    pred = -(2.0 * bts + sAs);
  }

  /* outputs: full trajectory with parameter rows re-inserted (optas/solver.py:139-157) */
  if (Q_out) {
    for (int i = 0; i < d->ndof; ++i)
      for (int t = 0; t < T; ++t) Q_out[(size_t)i * T + t] = in->Qfull[(size_t)i * T + t];
    for (int j = 0; j < n; ++j)
      for (int t = 0; t < T; ++t) Q_out[(size_t)d->opt_index[j] * T + t] = w.Q[(size_t)j * T + t];
  }
  if (dQ_out) {
    memset(dQ_out, 0, sizeof(double) * d->ndof * (T - 1));
    for (int j = 0; j < n; ++j)
      for (int t = 1; t < T - 1; ++t)
        dQ_out[(size_t)d->opt_index[j] * (T - 1) + t] = (w.Q[(size_t)j * T + t + 1] - w.Q[(size_t)j * T + t]) / dt;
  }
  if (cost_out) *cost_out = f;
  if (iters_out) *iters_out = k;
  if (status_out) *status_out = status;

  eval_free(&w.cur);
  eval_free(&w.tri);
  free(w.Q);
  free(w.Qtry);
  free(w.D);
  free(w.Aun);
  free(w.e);
  free(w.rhs);
  free(w.dx);
  free(w.L);
  free(w.C);
  free(w.bfull);
  free(w.act);
}

/* ------------------------------------------------------------------ batch entry points */
typedef struct orc_scene {
  const float* c_all;
  const float* c_obs;
  int32_t shape[3];
  double origin[3];
  double res;
} orc_scene;

static void make_instance(orc_instance* in, const gto_robot_desc* d, const gto_solver_opts* o,
                          const orc_scene* sc, int n_max, int b, const double* goals,
                          const int32_t* n_goals, const double* standoff, const double* base_pos,
                          const double* Qfull) {
  in->d = d;
  in->o = o;
  in->field.c_all = sc->c_all;
  in->field.c_obs = sc->c_obs ? sc->c_obs : sc->c_all;
  memcpy(in->field.shape, sc->shape, sizeof sc->shape);
  memcpy(in->field.origin, sc->origin, sizeof sc->origin);
  in->field.res = sc->res;
  in->n_goals = n_goals ? n_goals[b] : 0;
  in->goals = goals ? goals + (size_t)b * n_max * 16 : NULL;
  in->standoff = standoff ? standoff + (size_t)b * 16 : NULL;
  memcpy(in->base, base_pos + 3 * (size_t)b, sizeof in->base);
  in->Qfull = Qfull + (size_t)b * d->ndof * o->T;
}

/* Same contract as gto_solve_batch; `scenes` is indexed by scene_id[b]. f_trace [B][max_iter+1] or NULL. */
int orc_solve_batch(const gto_robot_desc* d, const gto_solver_opts* o, const orc_scene* scenes,
                    int32_t B, int32_t n_max, const int32_t* scene_id, const double* qc,
                    const double* goals, const int32_t* n_goals, const double* standoff,
                    const double* base_pos, const double* Q0, double* Q_out, double* dQ_out,
                    double* cost_out, int32_t* iters_out, int32_t* status_out, double* f_trace,
                    int32_t n_threads) {
  if (d->n_opt > NMAX || d->n_frames > GTO_MAX_FRAMES || d->n_links > GTO_MAX_LINKS || d->ndof > GTO_MAX_DOF)
    return GTO_ERR_UNSUPPORTED;
  const int T = o->T;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b) {
    orc_instance in;
    make_instance(&in, d, o, scenes + scene_id[b], n_max, b, goals, n_goals, standoff, base_pos, Q0);
    solve_instance(&in, qc + (size_t)b * d->ndof, Q_out ? Q_out + (size_t)b * d->ndof * T : NULL,
                   dQ_out ? dQ_out + (size_t)b * d->ndof * (T - 1) : NULL, cost_out ? cost_out + b : NULL,
                   iters_out ? iters_out + b : NULL, status_out ? status_out + b : NULL,
                   f_trace ? f_trace + (size_t)b * (o->max_iter + 1) : NULL);
  }
  return GTO_OK;
}

/* ------------------------------------------------------------------ inverse kinematics
 * gto/ik_solver.py:30-110 — the pre-step that produces q_solutions for plan_goalset: T = 1,
 *   min_q  sum_k || T_g(q) p_k - RT G p_k ||^2  +  w_obstacle * sum_pts c_obs[off(x(q))]      (:47-70)
 *   s.t.   lo <= q <= hi on the optimised joints                                               (:73)
 * with p_k the gripper surface points, G = gripper_tf (link_gripper in the link_ee frame), the collision
 * term the PLAIN sum of the cost (not squared, :70) over every collision link, the field shifted by
 * base_position.  Same projected Levenberg-Marquardt as the trajectory solve, one dense block: the pose
 * term is least squares (A = J^T J, b = J^T r); the collision term has no curvature model and enters the
 * gradient only, 2 b = grad f, i.e. b += (w/2) sum J_pt^T grad c (central differences,
 * gto/sdf_callback.py:90-114; zero in GTO_GRAD_ZERO mode, which is what CasADi AD sees). */
typedef struct ik_eval {
  double f_pos, f_obs;
  double A[NMAX * NMAX], b[NMAX];
} ik_eval;

static void ik_evaluate(const gto_robot_desc* d, const gto_solver_opts* o, const orc_field* fld, const double* base,
                        const double* q, const double* RT16, int want_deriv, ik_eval* ev) {
  const int n = d->n_opt;
  orc_kin* k = (orc_kin*)malloc(sizeof(orc_kin));
  kin_compute(d, q, k);
  memset(ev->A, 0, sizeof ev->A);
  memset(ev->b, 0, sizeof ev->b);
  /* pose term: gripper points against their goal positions */
  const double* Ag = k->frames + 12 * d->frame_gripper;
  const unsigned ancg = k->anc[d->frame_gripper];
  double Y[12];
  goal_target(k, d, RT16, NULL, Y);
  double fp = 0.0;
  for (int p = 0; p < d->n_gripper_points; ++p) {
    double xa[3], xb[3], r[3];
    aff_apply(Ag, d->gripper_points + 3 * p, xa);
    aff_apply(Y, d->gripper_points + 3 * p, xb);
    for (int c = 0; c < 3; ++c) {
      r[c] = xa[c] - xb[c];
      fp += r[c] * r[c];
    }
    if (want_deriv) {
      double J[3][NMAX];
      for (int j = 0; j < n; ++j) {
        double col[3] = {0, 0, 0};
        if (ancg >> j & 1u) point_jac_col(k, j, xa, col);
        J[0][j] = col[0];
        J[1][j] = col[1];
        J[2][j] = col[2];
      }
      for (int i = 0; i < n; ++i) {
        ev->b[i] += J[0][i] * r[0] + J[1][i] * r[1] + J[2][i] * r[2];
        for (int j = 0; j < n; ++j) ev->A[i * n + j] += J[0][i] * J[0][j] + J[1][i] * J[1][j] + J[2][i] * J[2][j];
      }
    }
  }
  ev->f_pos = fp;
  /* collision term */
  double fo = 0.0;
  if (fld) {
    double gsum[NMAX];
    for (int j = 0; j < n; ++j) gsum[j] = 0.0;
    for (int p = 0; p < d->n_points; ++p) {
      int l = d->point_link[p];
      double y[3], x[3];
      aff_apply(k->vis + 12 * l, d->points + 3 * p, y);
      for (int c = 0; c < 3; ++c) x[c] = y[c] + base[c];
      int32_t idx[3];
      voxel_index(fld, x, idx);
      fo += field_value(fld, fld->c_obs, idx);
      if (want_deriv && o->grad_mode == GTO_GRAD_CENTRAL_DIFF) {
        double g[3];
        field_grad(fld, fld->c_obs, idx, g);
        if (g[0] != 0.0 || g[1] != 0.0 || g[2] != 0.0) {
          unsigned anc = k->anc[d->link_frame[l]];
          for (int j = 0; j < n; ++j)
            if (anc >> j & 1u) {
              double col[3];
              point_jac_col(k, j, y, col);
              gsum[j] += g[0] * col[0] + g[1] * col[1] + g[2] * col[2];
            }
        }
      }
    }
    if (want_deriv)
      for (int j = 0; j < n; ++j) ev->b[j] += 0.5 * o->w_obstacle * gsum[j];
  }
  ev->f_obs = o->w_obstacle * fo;
  free(k);
}

static void solve_ik_instance(const gto_robot_desc* d, const gto_solver_opts* o, const orc_field* fld,
                              const double* base, const double* q0, const double* RT16, double* q_out,
                              double* cost_out, int32_t* iters_out, int32_t* status_out) {
  const int n = d->n_opt;
  double qtry[GTO_MAX_DOF], x[NMAX], xtry[NMAX];
  for (int i = 0; i < d->ndof; ++i) qtry[i] = q0[i];
  for (int j = 0; j < n; ++j) {
    double v = q0[d->opt_index[j]];
    if (v < d->lower[j]) v = d->lower[j];
    if (v > d->upper[j]) v = d->upper[j];
    xtry[j] = v;
  }
  ik_eval cur, tri;
  double lambda = o->lambda0, nu = 2.0, f = INFINITY, pred = 0.0;
  int status = GTO_STATUS_MAX_ITER, first = 1, k = 0;
  for (;; ++k) {
    for (int j = 0; j < n; ++j) qtry[d->opt_index[j]] = xtry[j];
    ik_evaluate(d, o, fld, base, qtry, RT16, 1, &tri);
    const double f_try = tri.f_pos + tri.f_obs;
    int done = 0;
    if (first) {
      first = 0;
      memcpy(x, xtry, sizeof x);
      f = f_try;
      cur = tri;
    } else if (f_try < f && pred > 0.0) {
      double df = f - f_try, rho = df / pred;
      memcpy(x, xtry, sizeof x);
      f = f_try;
      cur = tri;
      double s = 2.0 * rho - 1.0, fac = 1.0 - s * s * s;
      if (fac < 1.0 / 3.0) fac = 1.0 / 3.0;
      lambda *= fac;
      if (lambda < 1e-12) lambda = 1e-12;
      nu = 2.0;
      if (df <= o->tol_rel_f * (1.0 + f)) {
        status = GTO_STATUS_CONVERGED;
        done = 1;
      }
    } else {
      lambda *= nu;
      nu *= 2.0;
      if (lambda > 1e15) {
        status = GTO_STATUS_CONVERGED;
        done = 1;
      }
    }
    if (done) break;
    if (k >= o->max_iter) {
      status = GTO_STATUS_MAX_ITER;
      break;
    }
    int act[NMAX];
    double D[NMAX * NMAX], rhs[NMAX], dx[NMAX], e[NMAX], L[NMAX * NMAX], C[NMAX * NMAX];
    for (int i = 0; i < n; ++i) act[i] = (x[i] <= d->lower[i] && cur.b[i] > 0.0) || (x[i] >= d->upper[i] && cur.b[i] < 0.0);
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) {
        double v = cur.A[i * n + j];
        if (act[i] || act[j]) v = (i == j) ? 1.0 : 0.0;
        else if (i == j) v *= (1.0 + lambda);
        D[i * n + j] = v;
      }
      rhs[i] = act[i] ? 0.0 : -cur.b[i];
      e[i] = 0.0;
    }
    if (!block_tridiag_solve(1, n, D, e, rhs, dx, L, C)) {
      status = GTO_STATUS_NUMERICAL;
      break;
    }
    double maxstep = 0.0;
    for (int i = 0; i < n; ++i) {
      double v = x[i] + dx[i];
      if (v < d->lower[i]) v = d->lower[i];
      if (v > d->upper[i]) v = d->upper[i];
      xtry[i] = v;
      dx[i] = v - x[i];
      if (fabs(dx[i]) > maxstep) maxstep = fabs(dx[i]);
    }
    if (maxstep < o->tol_step) {
      status = GTO_STATUS_CONVERGED;
      break;
    }
    double bts = 0.0, sAs = 0.0;
    for (int i = 0; i < n; ++i) {
      bts += cur.b[i] * dx[i];
      double r = 0.0;
      for (int j = 0; j < n; ++j) r += cur.A[i * n + j] * dx[j];
      sAs += dx[i] * r;
    }
    pred = -(2.0 * bts + sAs);
  }
  for (int i = 0; i < d->ndof; ++i) q_out[i] = q0[i]; /* parameter joints as given (optas/solver.py:139-157) */
  for (int j = 0; j < n; ++j) q_out[d->opt_index[j]] = x[j];
  if (cost_out) *cost_out = f;
  if (iters_out) *iters_out = k;
  if (status_out) *status_out = status;
}

/* Same contract as gto_solve_ik_batch.  scene_id NULL: no collision term (collision_avoidance=False). */
int orc_solve_ik_batch(const gto_robot_desc* d, const gto_solver_opts* o, const orc_scene* scenes, int32_t B,
                       const int32_t* scene_id, const double* q0, const double* goals, const double* base_pos,
                       double* q_out, double* cost_out, int32_t* iters_out, int32_t* status_out, int32_t n_threads) {
  if (d->n_opt > NMAX || d->n_frames > GTO_MAX_FRAMES || d->n_links > GTO_MAX_LINKS || d->ndof > GTO_MAX_DOF)
    return GTO_ERR_UNSUPPORTED;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b) {
    orc_field fld;
    const orc_field* pf = NULL;
    double base[3] = {0, 0, 0};
    if (scene_id) {
      const orc_scene* sc = scenes + scene_id[b];
      fld.c_all = sc->c_all;
      fld.c_obs = sc->c_obs ? sc->c_obs : sc->c_all;
      memcpy(fld.shape, sc->shape, sizeof sc->shape);
      memcpy(fld.origin, sc->origin, sizeof sc->origin);
      fld.res = sc->res;
      pf = &fld;
      if (base_pos) memcpy(base, base_pos + 3 * (size_t)b, sizeof base);
    }
    solve_ik_instance(d, o, pf, base, q0 + (size_t)b * d->ndof, goals + (size_t)b * 16, q_out + (size_t)b * d->ndof,
                      cost_out ? cost_out + b : NULL, iters_out ? iters_out + b : NULL, status_out ? status_out + b : NULL);
  }
  return GTO_OK;
}

/* ------------------------------------------------------------------ base placement
 * gto/base_planner.py:35-94.  Unknowns z = [x, y, theta ; q_1 .. q_n] (one arm configuration per goal):
 *   f = w |(x,y,theta)|^2 + sum_i sum_k | A(q_i) p_k - (B(x,y,theta) RT_i G(q_i)) p_k |^2 ,
 * B = rt2tr(rotz(theta), [x,y,0]) (:49-51), A = gripper-link transform (:66-68), joint limits on every
 * q_i (:92) and -pi <= theta <= pi (:55).  The reference also carries n-1 unused copies of the task
 * state (T = goal_size, only column 0 enters the cost, :44-46); they stay at their zero seed and are
 * not represented here.  Same projected Levenberg-Marquardt rules as solve_ik_instance; the normal
 * equations are assembled point by point and solved by one dense Cholesky factorisation. */
static void base_evaluate(const gto_robot_desc* d, int ng, const double* z, const double* qc, const double* goals,
                          double w_effort, int want_deriv, double* f_out, double* A, double* b) {
  const int n = d->n_opt, N = 3 + n * ng;
  orc_kin* k = (orc_kin*)malloc(sizeof(orc_kin));
  if (want_deriv) {
    memset(A, 0, sizeof(double) * (size_t)N * N);
    memset(b, 0, sizeof(double) * (size_t)N);
  }
  const double th = z[2], cs = cos(th), sn = sin(th);
  const double Bm[12] = {cs, -sn, 0, z[0], sn, cs, 0, z[1], 0, 0, 1, 0};
  double f = w_effort * (z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  if (want_deriv)
    for (int a = 0; a < 3; ++a) {
      A[a * N + a] += w_effort;
      b[a] += w_effort * z[a];
    }
  double q[GTO_MAX_DOF];
  for (int i = 0; i < ng; ++i) {
    for (int c = 0; c < d->ndof; ++c) q[c] = qc[c];
    for (int j = 0; j < n; ++j) q[d->opt_index[j]] = z[3 + i * n + j];
    kin_compute(d, q, k);
    const double* Ag = k->frames + 12 * d->frame_gripper;
    const unsigned ancg = k->anc[d->frame_gripper];
    double Y0[12];
    goal_target(k, d, goals + 16 * (size_t)i, NULL, Y0);
    for (int p = 0; p < d->n_gripper_points; ++p) {
      double xa[3], g[3], tau[3], r[3];
      aff_apply(Ag, d->gripper_points + 3 * p, xa);
      aff_apply(Y0, d->gripper_points + 3 * p, g);
      aff_apply(Bm, g, tau);
      for (int c = 0; c < 3; ++c) {
        r[c] = xa[c] - tau[c];
        f += r[c] * r[c];
      }
      if (!want_deriv) continue;
      /* residual Jacobian columns: base x, y, theta, then this goal's joints */
      double J[3][3 + NMAX];
      J[0][0] = -1.0, J[1][0] = 0.0, J[2][0] = 0.0;
      J[0][1] = 0.0, J[1][1] = -1.0, J[2][1] = 0.0;
      J[0][2] = tau[1] - z[1], J[1][2] = -(tau[0] - z[0]), J[2][2] = 0.0;
      for (int j = 0; j < n; ++j) {
        double col[3] = {0, 0, 0};
        if (ancg >> j & 1u) point_jac_col(k, j, xa, col);
        J[0][3 + j] = col[0], J[1][3 + j] = col[1], J[2][3 + j] = col[2];
      }
      for (int a = 0; a < 3 + n; ++a) {
        const int ia = a < 3 ? a : 3 + i * n + (a - 3);
        b[ia] += J[0][a] * r[0] + J[1][a] * r[1] + J[2][a] * r[2];
        for (int c = 0; c < 3 + n; ++c) {
          const int ic = c < 3 ? c : 3 + i * n + (c - 3);
          A[ia * N + ic] += J[0][a] * J[0][c] + J[1][a] * J[1][c] + J[2][a] * J[2][c];
        }
      }
    }
  }
  *f_out = f;
  free(k);
}

static void solve_base_instance(const gto_robot_desc* d, const gto_solver_opts* o, int ng, int n_max, const double* qc,
                                const double* goals, double w_effort, int max_iter, double* y_out, double* q_out,
                                double* cost_out, int32_t* iters_out, int32_t* status_out) {
  const int n = d->n_opt, N = 3 + n * ng;
  double* x = (double*)malloc(sizeof(double) * N * 8);
  double *xtry = x + N, *lo = x + 2 * N, *hi = x + 3 * N, *bcur = x + 4 * N, *btry = x + 5 * N, *rhs = x + 6 * N,
         *dx = x + 7 * N;
  double* Abuf = (double*)malloc(sizeof(double) * (size_t)N * N * 3);
  double *Acur = Abuf, *Atry = Abuf + (size_t)N * N, *M = Abuf + 2 * (size_t)N * N;
  int* act = (int*)malloc(sizeof(int) * N);
  lo[0] = lo[1] = -INFINITY, hi[0] = hi[1] = INFINITY;
  lo[2] = -3.141592653589793, hi[2] = 3.141592653589793; /* np.pi, gto/base_planner.py:55 */
  xtry[0] = xtry[1] = xtry[2] = 0.0; /* gto/base_planner.py:105 */
  for (int i = 0; i < ng; ++i)
    for (int j = 0; j < n; ++j) {
      double v = qc[d->opt_index[j]]; /* every goal starts from the current configuration (:104) */
      lo[3 + i * n + j] = d->lower[j], hi[3 + i * n + j] = d->upper[j];
      if (v < d->lower[j]) v = d->lower[j];
      if (v > d->upper[j]) v = d->upper[j];
      xtry[3 + i * n + j] = v;
    }
  double lambda = o->lambda0, nu = 2.0, f = INFINITY, pred = 0.0;
  int status = GTO_STATUS_MAX_ITER, first = 1, k = 0;
  for (;; ++k) {
    double f_try;
    base_evaluate(d, ng, xtry, qc, goals, w_effort, 1, &f_try, Atry, btry);
    int done = 0, take = 0;
    if (first) {
      first = 0;
      take = 1;
    } else if (f_try < f && pred > 0.0) {
      double df = f - f_try, rho = df / pred;
      take = 1;
      double s = 2.0 * rho - 1.0, fac = 1.0 - s * s * s;
      if (fac < 1.0 / 3.0) fac = 1.0 / 3.0;
      lambda *= fac;
      if (lambda < 1e-12) lambda = 1e-12;
      nu = 2.0;
      if (df <= o->tol_rel_f * (1.0 + f_try)) {
        status = GTO_STATUS_CONVERGED;
        done = 1;
      }
    } else {
      lambda *= nu;
      nu *= 2.0;
      if (lambda > 1e15) {
        status = GTO_STATUS_CONVERGED;
        done = 1;
      }
    }
    if (take) {
      memcpy(x, xtry, sizeof(double) * N);
      memcpy(bcur, btry, sizeof(double) * N);
      double* t = Acur;
      Acur = Atry;
      Atry = t;
      f = f_try;
    }
    if (done) break;
    if (k >= max_iter) {
      status = GTO_STATUS_MAX_ITER;
      break;
    }
    for (int i = 0; i < N; ++i) act[i] = (x[i] <= lo[i] && bcur[i] > 0.0) || (x[i] >= hi[i] && bcur[i] < 0.0);
    for (int i = 0; i < N; ++i) {
      for (int j = 0; j < N; ++j) {
        double v = Acur[(size_t)i * N + j];
        if (act[i] || act[j]) v = (i == j) ? 1.0 : 0.0;
        else if (i == j) v *= (1.0 + lambda);
        M[(size_t)i * N + j] = v;
      }
      rhs[i] = act[i] ? 0.0 : -bcur[i];
    }
    if (!chol(M, N)) {
      status = GTO_STATUS_NUMERICAL;
      break;
    }
    for (int i = 0; i < N; ++i) {
      double s = rhs[i];
      for (int c = 0; c < i; ++c) s -= M[(size_t)i * N + c] * dx[c];
      dx[i] = s / M[(size_t)i * N + i];
    }
    for (int i = N - 1; i >= 0; --i) {
      double s = dx[i];
      for (int c = i + 1; c < N; ++c) s -= M[(size_t)c * N + i] * dx[c];
      dx[i] = s / M[(size_t)i * N + i];
    }
    double maxstep = 0.0;
    for (int i = 0; i < N; ++i) {
      double v = x[i] + dx[i];
      if (v < lo[i]) v = lo[i];
      if (v > hi[i]) v = hi[i];
      xtry[i] = v;
      dx[i] = v - x[i];
      if (fabs(dx[i]) > maxstep) maxstep = fabs(dx[i]);
    }
    if (maxstep < o->tol_step) {
      status = GTO_STATUS_CONVERGED;
      break;
    }
    double bts = 0.0, sAs = 0.0;
    for (int i = 0; i < N; ++i) {
      bts += bcur[i] * dx[i];
      double r = 0.0;
      for (int j = 0; j < N; ++j) r += Acur[(size_t)i * N + j] * dx[j];
      sAs += dx[i] * r;
    }
    pred = -(2.0 * bts + sAs);
  }
  for (int a = 0; a < 3; ++a) y_out[a] = x[a];
  for (int i = 0; i < n_max; ++i) { /* parameter joints as given (optas/solver.py:139-157); unused rows = qc */
    for (int c = 0; c < d->ndof; ++c) q_out[(size_t)i * d->ndof + c] = qc[c];
    if (i < ng)
      for (int j = 0; j < n; ++j) q_out[(size_t)i * d->ndof + d->opt_index[j]] = x[3 + i * n + j];
  }
  if (cost_out) *cost_out = f;
  if (iters_out) *iters_out = k;
  if (status_out) *status_out = status;
  free(act);
  free(Abuf);
  free(x);
}

/* Same contract as gto_solve_base_batch. */
int orc_solve_base_batch(const gto_robot_desc* d, const gto_solver_opts* o, int32_t B, int32_t n_max,
                         const int32_t* n_goals, const double* qc, const double* goals, double w_effort,
                         int32_t max_iter, double* y_out, double* q_out, double* cost_out, int32_t* iters_out,
                         int32_t* status_out, int32_t n_threads) {
  if (d->n_opt > NMAX || d->n_frames > GTO_MAX_FRAMES || d->n_links > GTO_MAX_LINKS || d->ndof > GTO_MAX_DOF)
    return GTO_ERR_UNSUPPORTED;
  for (int b = 0; b < B; ++b)
    if (n_goals[b] < 1 || n_goals[b] > n_max) return GTO_ERR_INVALID_ARG;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b)
    solve_base_instance(d, o, n_goals[b], n_max, qc + (size_t)b * d->ndof, goals + (size_t)b * n_max * 16, w_effort, max_iter,
                        y_out + 3 * (size_t)b, q_out + (size_t)b * n_max * d->ndof, cost_out ? cost_out + b : NULL,
                        iters_out ? iters_out + b : NULL, status_out ? status_out + b : NULL);
  return GTO_OK;
}

/* Base-placement objective at a given point (same contract as gto_eval_base_objective): y [B][3],
 * q [B][n_max][ndof] one arm configuration per goal (parameter joints taken from row 0), goals [B][n_max][16]. */
int orc_eval_base_objective(const gto_robot_desc* d, int32_t B, int32_t n_max, const int32_t* n_goals, const double* y,
                            const double* q, const double* goals, double w_effort, double* cost_out) {
  if (d->n_opt > NMAX || n_max < 1) return GTO_ERR_UNSUPPORTED;
  const int n = d->n_opt;
  double* z = (double*)malloc(sizeof(double) * (3 + (size_t)n * n_max));
  for (int b = 0; b < B; ++b) {
    const double* qb = q + (size_t)b * n_max * d->ndof;
    for (int a = 0; a < 3; ++a) z[a] = y[3 * (size_t)b + a];
    for (int i = 0; i < n_goals[b]; ++i)
      for (int j = 0; j < n; ++j) z[3 + i * n + j] = qb[(size_t)i * d->ndof + d->opt_index[j]];
    base_evaluate(d, n_goals[b], z, qb, goals + (size_t)b * n_max * 16, w_effort, 0, cost_out + b, NULL, NULL);
  }
  free(z);
  return GTO_OK;
}

/* Objective terms at given trajectories (same contract as gto_eval_objective). */
int orc_eval_objective(const gto_robot_desc* d, const gto_solver_opts* o, const orc_scene* scenes,
                       int32_t B, int32_t n_max, const int32_t* scene_id, const double* goals,
                       const int32_t* n_goals, const double* standoff, const double* base_pos,
                       const double* Q, double* f_goal, double* f_obs, double* f_vel, int32_t* argmin) {
  const int T = o->T, n = d->n_opt;
  for (int b = 0; b < B; ++b) {
    orc_instance in;
    make_instance(&in, d, o, scenes + scene_id[b], n_max, b, goals, n_goals, standoff, base_pos, Q);
    orc_eval ev;
    eval_alloc(&ev, T, n);
    double* Qopt = (double*)malloc(sizeof(double) * n * T);
    for (int j = 0; j < n; ++j)
      memcpy(Qopt + (size_t)j * T, in.Qfull + (size_t)d->opt_index[j] * T, sizeof(double) * T);
    evaluate(&in, Qopt, 0, &ev);
    if (f_goal) f_goal[b] = ev.f_goal;
    if (f_obs) f_obs[b] = ev.f_obs;
    if (f_vel) f_vel[b] = ev.f_vel;
    if (argmin) argmin[b] = ev.goal_argmin;
    free(Qopt);
    eval_free(&ev);
  }
  return GTO_OK;
}

/* Gauss-Newton blocks (same contract as gto_eval_obstacle_normal_eq) plus the goal blocks. */
int orc_eval_normal_eq(const gto_robot_desc* d, const gto_solver_opts* o, const orc_scene* scenes,
                       int32_t B, int32_t n_max, const int32_t* scene_id, const double* goals,
                       const int32_t* n_goals, const double* standoff, const double* base_pos,
                       const double* Q, double* JtJ, double* Jtr, double* sumsq,
                       double* JtJ_goal /*[B][2][n][n]*/, double* Jtr_goal /*[B][2][n]*/) {
  const int T = o->T, n = d->n_opt;
  for (int b = 0; b < B; ++b) {
    orc_instance in;
    make_instance(&in, d, o, scenes + scene_id[b], n_max, b, goals, n_goals, standoff, base_pos, Q);
    double dummy_goal[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (!goals) {
      in.n_goals = 1;
      in.goals = dummy_goal;
    }
    orc_eval ev;
    eval_alloc(&ev, T, n);
    double* Qopt = (double*)malloc(sizeof(double) * n * T);
    for (int j = 0; j < n; ++j)
      memcpy(Qopt + (size_t)j * T, in.Qfull + (size_t)d->opt_index[j] * T, sizeof(double) * T);
    evaluate(&in, Qopt, 1, &ev);
    if (JtJ) memcpy(JtJ + (size_t)b * T * n * n, ev.JtJ_obs, sizeof(double) * T * n * n);
    if (Jtr) memcpy(Jtr + (size_t)b * T * n, ev.Jtr_obs, sizeof(double) * T * n);
    if (sumsq) memcpy(sumsq + (size_t)b * T, ev.sumsq, sizeof(double) * T);
    if (JtJ_goal)
      for (int w = 0; w < 2; ++w)
        for (int i = 0; i < n * n; ++i) JtJ_goal[((size_t)b * 2 + w) * n * n + i] = ev.JtJ_goal[w][i];
    if (Jtr_goal)
      for (int w = 0; w < 2; ++w)
        for (int i = 0; i < n; ++i) Jtr_goal[((size_t)b * 2 + w) * n + i] = ev.Jtr_goal[w][i];
    free(Qopt);
    eval_free(&ev);
  }
  return GTO_OK;
}

/* World surface points, offsets, values, gradients (same contract as gto_eval_points). */
int orc_eval_points(const gto_robot_desc* d, const orc_scene* sc, int32_t nq, const double* q,
                    const double* base_pos, int32_t use_obs, double* xyz_out, int32_t* offset_out,
                    double* value_out, double* grad_out) {
  orc_kin* k = (orc_kin*)malloc(sizeof(orc_kin));
  orc_field f;
  f.c_all = sc->c_all;
  f.c_obs = sc->c_obs ? sc->c_obs : sc->c_all;
  memcpy(f.shape, sc->shape, sizeof f.shape);
  memcpy(f.origin, sc->origin, sizeof f.origin);
  f.res = sc->res;
  const float* data = use_obs ? f.c_obs : f.c_all;
  for (int i = 0; i < nq; ++i) {
    kin_compute(d, q + (size_t)i * d->ndof, k);
    for (int p = 0; p < d->n_points; ++p) {
      double y[3], x[3];
      aff_apply(k->vis + 12 * d->point_link[p], d->points + 3 * p, y);
      for (int r = 0; r < 3; ++r) x[r] = y[r] + base_pos[3 * i + r];
      size_t o = (size_t)i * d->n_points + p;
      if (xyz_out) memcpy(xyz_out + 3 * o, x, sizeof x);
      if (sc->c_all) {
        int32_t idx[3];
        voxel_index(&f, x, idx);
        if (offset_out) offset_out[o] = (int32_t)flat_offset(&f, idx[0], idx[1], idx[2]);
        if (value_out) value_out[o] = field_value(&f, data, idx);
        if (grad_out) field_grad(&f, data, idx, grad_out + 3 * o);
      }
    }
  }
  free(k);
  return GTO_OK;
}

/* gto/gto_models.py:204-215 compute_plan_cost: plain sum of c_obs over waypoints and points. */
int orc_plan_cost(const gto_robot_desc* d, const orc_scene* sc, int32_t n, int32_t T, const double* plans,
                  const double* base_pos, double* cost_out, double* dist_out) {
  orc_kin* k = (orc_kin*)malloc(sizeof(orc_kin));
  orc_field f;
  f.c_all = sc->c_all;
  f.c_obs = sc->c_obs ? sc->c_obs : sc->c_all;
  memcpy(f.shape, sc->shape, sizeof f.shape);
  memcpy(f.origin, sc->origin, sizeof f.origin);
  f.res = sc->res;
  double q[GTO_MAX_DOF];
  for (int i = 0; i < n; ++i) {
    const double* plan = plans + (size_t)i * d->ndof * T;
    double cost = 0.0;
    for (int t = 0; t < T; ++t) {
      for (int j = 0; j < d->ndof; ++j) q[j] = plan[(size_t)j * T + t];
      kin_compute(d, q, k);
      for (int p = 0; p < d->n_points; ++p) {
        double y[3], x[3];
        aff_apply(k->vis + 12 * d->point_link[p], d->points + 3 * p, y);
        for (int r = 0; r < 3; ++r) x[r] = y[r] + base_pos[r];
        int32_t idx[3];
        voxel_index(&f, x, idx);
        cost += field_value(&f, f.c_obs, idx);
      }
    }
    double dd = 0.0;
    for (int j = 0; j < d->ndof; ++j) {
      double v = plan[(size_t)j * T] - plan[(size_t)j * T + T - 1];
      dd += v * v;
    }
    cost_out[i] = cost;
    dist_out[i] = sqrt(dd);
  }
  free(k);
  return GTO_OK;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
