// gto_device.h — device-side data layout and small math shared by the GTO kernels (gfx950).
//
// Data layout in HBM (all FP64 unless stated; "n" = optimised joints, "L" = collision links):
//   RobotDev            one per handle, read through scalar loads (uniform addresses)
//   points              SoA px[], py[], pz[], sorted by collision link; chunk table {link,start,count<=64}
//                       so that one wavefront processes one link-uniform chunk per step
//   scene fields        float32 [nx*ny*nz] C order (x slowest), c_all and c_obs per scene
//   per-instance state  SoA over the batch: Q [B][n][T], kinematics handed from the step kernel to
//                       the obstacle kernel as visual transforms [B][T][L][12] and joint screws
//                       [B][T][n][6]; Gauss-Newton blocks [2 slots][B][T][n*n + n + 1]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gto_solver.h"

#define GTO_NB 8            // padded block size of the per-waypoint normal-equation blocks
// Broad phase: a chunk whose bounding sphere has radius rho voxels cannot put a point more than ceil(rho) voxel indices
// away from its centre's voxel on any axis (floor(x + t) <= floor(x) + ceil(t)), so it is culled when the Chebyshev
// distance from the centre's voxel to the nearest non-zero record exceeds R = ceil(rho + 1e-6) + GTO_BROAD_MARGIN.  The
// margin was 2 until round 3 ("floor of the centre, index rounding": both are inside the ceil already).
#ifndef GTO_BROAD_MARGIN
#define GTO_BROAD_MARGIN 0
#endif
#define GTO_GRAM 28         // 21 (6x6 symmetric wrench Gram) + 6 (c * wrench) + 1 (c^2)
#define GTO_WAVE 64
// LDS bank layout of the forward kinematics (fk_mfma_tree).  A wavefront works on four frames at once, one per 16-lane
// block of the matrix-core instruction, and every lane reads or writes ONE entry of its frame's 4x4 matrix (entry
// e = 4 row + col).  The LDS serves an 8-byte access 16 lanes at a time (32 banks x 4 B): sixteen consecutive lanes touch
// one ROW of four matrices, or -- the transposed read that turns a result into an A operand -- one COLUMN.  With 16
// doubles per frame every frame starts on bank 0 and both patterns are four-way conflicts (97 % of the obstacle kernel's
// bank-conflict cycles were here).  Frame f therefore keeps entry e at e ^ 5 (f & 3): the row r of the four frames of a
// wave then lands on four different quarters of the banks, and so does the column c (XOR with 0, 5, 10, 15 moves both the
// row bits and the column bits of e).  The operand tables (per frame O^T, c0, c1, K; per link Vo; per joint U) use the
// same placement, and hold the origin transposed so that all their reads are row reads.
#define GTO_FK_STRIDE 64
__host__ __device__ inline int fkx(int f, int e) { return 16 * f + (e ^ (5 * (f & 3))); }

struct RobotDev {
  int32_t n_frames, ndof, n_opt, n_links, n_points, n_chunks, n_gripper_points;
  int32_t frame_ee, frame_gripper;
  int32_t fk_rounds;  // ceil(log2(depth of the kinematic tree)): pointer-jumping rounds of the parallel FK
  int32_t opt_frame[GTO_MAX_OPT];  // frame whose joint is optimised joint j
  // operands of fk_mfma_tree, 16 doubles (entry e = 4*row + col of a 4x4 homogeneous matrix) each:
  // per frame the origin O, c0 = h + u u^T, c1 = delta - u u^T, K = [u]x (prismatic: the axis in the
  // translation column); then per link its visual origin; then per optimised joint U = [u;0 | e4]
  // ... then link_frame [L], opt_frame [n], prismatic flag [n], parent [F] as doubles (packed for the actual F, L, n)
  double fk_tab[(GTO_FK_STRIDE + 1) * GTO_MAX_FRAMES + 17 * GTO_MAX_LINKS + 18 * GTO_MAX_OPT];
  // The same table for the COMPACT tree the obstacle kernel walks (gto_api.hip, build_fk_tables): frames with fixed joints
  // are folded into the origins of the moving frames below them and into the visual origins of their links (a "world"
  // frame is appended when a link hangs on fixed frames only), so the pointer jumping has fewer frames and fewer levels
  // (Panda: 12 frames / 4 rounds -> 10 / 3).  Link transforms and joint screws are the same up to the order of the
  // products; the other kernels, which also read frames, keep the full table.
  double fk_tab_c[(GTO_FK_STRIDE + 1) * GTO_MAX_FRAMES + 17 * GTO_MAX_LINKS + 18 * GTO_MAX_OPT];
  int32_t n_cframes, fk_rounds_c;
  int32_t pb_ctl[GTO_MAX_FRAMES];   // control words of the step kernel's serial walk over the tree (gto_kernels.h, GTO_PB_SRC ...)
  double pb_eps;                     // widening of its culling radius, metres (single-precision storage of the transforms)
  int32_t pb_par[GTO_MAX_FRAMES];    // slot of the frame's joint among the actuated joints that are not optimised, or -1
  int32_t pb_parf[GTO_MAX_FRAMES];   // frame of such a slot
  int32_t pb_npar;
  int32_t cf_orig[GTO_MAX_FRAMES];  // frame whose joint value a compact frame takes (-1: the world frame)
  int32_t cf_type[GTO_MAX_FRAMES];  // its joint type
  int32_t parent[GTO_MAX_FRAMES];
  int32_t joint_type[GTO_MAX_FRAMES];
  int32_t q_index[GTO_MAX_FRAMES];
  int32_t opt_of_frame[GTO_MAX_FRAMES];  // optimised-joint slot driven by this frame's joint, or -1
  int32_t link_of_frame[GTO_MAX_FRAMES]; // collision link attached to this frame, or -1
  // serial FK walk of k_traj_solve: frames whose transform a later, non-adjacent child needs are parked in LDS
  int32_t xst_slot[GTO_MAX_FRAMES];      // parking slot of this frame's transform, or -1
  int32_t n_xst;
  uint32_t frame_free_mask;  // bit f: frame f is not the frame of an optimised joint (its joint value never changes in a solve)
  int32_t opt_of_dof[GTO_MAX_DOF];       // optimised-joint slot of actuated joint i, or -1 (parameter joint)
  uint32_t frame_anc[GTO_MAX_FRAMES];    // bit j: optimised joint j moves this frame
  double origin[GTO_MAX_FRAMES][12];     // rt2tr(rpy2r(rpy), xyz)  (optas/models.py:848-857)
  double axis_unit[GTO_MAX_FRAMES][3];   // unit(axis)              (optas/models.py:653-659)
  int32_t link_frame[GTO_MAX_LINKS];
  uint32_t link_anc[GTO_MAX_LINKS];
  // per entry of a waypoint's normal-equation block, the links that contribute to it: entry (i, j) of J^T J -> links both
  // joints sit above, entry i of J^T r (stored behind the NP x NP entries) -> links joint i sits above; [0]: NP = 8, [1]: 16
  uint32_t entry_links[2][16 * 16 + 16];
  double vis_origin[GTO_MAX_LINKS][12];  // rt2tr(rpy2r(vis_rpy), vis_xyz) (gto/gto_models.py:95-96)
  int32_t opt_index[GTO_MAX_OPT];
  double lower[GTO_MAX_OPT], upper[GTO_MAX_OPT];
  // reach[j]: upper bound on |dx/dq_j| over every surface point and configuration (metres per radian,
  // or 1 for a prismatic joint); < 0 disables temporal culling
  double reach[GTO_MAX_OPT];
  // reach_link[l][j]: the same bound for the points of link l alone (0 if joint j does not move it): what a step dq moves
  // a bounding sphere of link l by, at most (emptiness certificates, k_certify in gto_kernels.h)
  double reach_link[GTO_MAX_LINKS][GTO_MAX_OPT];
  // moments of the gripper point cloud p_k (gto/gto_planner.py:37): K, mu = sum p, M = sum p p^T
  double grip_count, grip_mu[3], grip_M[9];
};

// One voxel of the gather-friendly field layout built at gto_set_scene: the nearest-voxel cost and the
// three central differences c[i+e_a] - c[i-e_a] (clipped neighbours, exact in FP64), 32 B, so the hot
// loop fetches ONE record (two 16-B loads, one cache line) instead of seven scattered floats.
struct __attribute__((aligned(32))) VoxelRec {
  double dx, dy, dz;
  float c;
  float pad;
};

struct SceneDev {
  const float* c_all;
  const float* c_obs;
  const VoxelRec* r_all;
  const VoxelRec* r_obs;
  // Chebyshev distance (in voxels, capped) from every voxel to the nearest voxel whose record is
  // non-zero: lets a whole chunk of surface points be culled with one lookup (broad phase)
  const uint8_t* d_all;
  const uint8_t* d_obs;
  int32_t nx, ny, nz, valid;
  double ox, oy, oz, res, rinv, inv2r;
};

struct Chunk {
  int32_t link, start, count, pad;
  double cx, cy, cz, r;  // bounding sphere of the chunk's points in the link's visual-mesh frame
};

// ---------------------------------------------------------------- 3x4 affine helpers (row-major)
__host__ __device__ inline void aff_mul(const double* a, const double* b, double* c) {
  double r[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j] + a[4 * i + 2] * b[8 + j];
      if (j == 3) s += a[4 * i + 3];
      r[4 * i + j] = s;
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) c[i] = r[i];
}
__host__ __device__ inline void mat3_mul(const double* a, const double* b, double* c) {
  double r[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) c[i] = r[i];
}
// optas/spatialmath.py:186-211 rpy2r 'zyx' = rotz(y) @ roty(p) @ rotx(r)
__host__ __device__ inline void rpy2r(const double* rpy, double* R) {
  double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
  double Rz[9] = {cy, -sy, 0, sy, cy, 0, 0, 0, 1};
  double Ry[9] = {cp, 0, sp, 0, 1, 0, -sp, 0, cp};
  double Rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr};
  double t[9];
  mat3_mul(Rz, Ry, t);
  mat3_mul(t, Rx, R);
}
__host__ __device__ inline void rt2aff(const double* R, const double* t, double* m) {
  for (int i = 0; i < 3; ++i) {
    m[4 * i] = R[3 * i];
    m[4 * i + 1] = R[3 * i + 1];
    m[4 * i + 2] = R[3 * i + 2];
    m[4 * i + 3] = t[i];
  }
}
__host__ __device__ inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// index of (i,j), i<=j, in the packed upper triangle of a symmetric 6x6
__host__ __device__ inline int sym6(int i, int j) {
  if (i > j) {
    int t = i;
    i = j;
    j = t;
  }
  return i * 6 - (i * (i - 1)) / 2 + (j - i);
}
