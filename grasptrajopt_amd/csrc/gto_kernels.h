// gto_kernels.h — HIP kernels of the GTO inner solver for CDNA4 (gfx950, wave64).
//
//   k_obstacle_gram   the dominant kernel: one workgroup per (slot, group of three waypoints).  Prologue: joint
//                     values of the slot's trial -> forward kinematics as a parallel prefix over the kinematic
//                     tree on the FP64 matrix cores (fk_mfma_tree).  Broad phase: bounding spheres of the 64-point
//                     chunks against a Chebyshev distance field.  Loop over the surviving chunks: visual transform
//                     from LDS -> world point -> exact voxel index -> ONE 32-B record (nearest-voxel cost + the three
//                     central differences) -> wrench (y x grad, grad) appended to a per-wave list -> 6x6 Gram per
//                     (waypoint, link) folded with v_mfma_f64_16x16x4.  Epilogue: projection onto the joint screws
//                     -> J^T J, J^T r, sum c^2 per waypoint.  Extra workgroups: goal-set and velocity terms.
//   k_lm_init/k_lm_step  one workgroup (four wavefronts) per instance / slot: accept/reject, bound active set,
//                     block-tridiagonal solve from both ends, projected step, new trial; hands a finished instance's
//                     slot to the next one of the call
//   k_lm_finalize     assemble Q / dQ / cost / status
//   k_ik_solve, k_base_solve   whole Levenberg-Marquardt loops in one workgroup (IK pre-filter, base placement)
//   k_depth_backproject, k_depth_sdf   cost field from a depth image
// Bound: latency and instruction issue, not HBM or MFMA (DESIGN.md sections 5-6): the field of a scene is
// L2-resident and the broad phase skips 97 % of the chunks; SURVEY.md 8d's algorithmic 28 B per point-waypoint
// are what bench.py prices the kernel against.
#pragma once
#include "gto_device.h"

// doubles per (instance, waypoint) block: NP x NP J^T J (row-major), NP J^T r, sum c^2, pad.  NP = 8 for robots with up
// to eight optimised joints (Panda, Fetch arm: the tuned path), 16 beyond that (mobile manipulators)
template <int NP>
struct Blk {
  static constexpr int JTJ = 0, JTR = NP * NP, SS = NP * NP + NP, STRIDE = NP * NP + NP + 8;
};
#define BLK_JTJ 0
#define BLK_JTR 64
#define BLK_SS 72
#define BLK_STRIDE 80  // = Blk<8>: the kernels that only exist for NP = 8 (k_lm_step, k_ik_solve, k_base_solve, k_traj_solve)
#define GTO_MAX_ACTIVE 256   // chunks per robot (16 K surface points)
#define GTO_MAX_TG 8         // waypoints per workgroup of the obstacle kernel
#define GTO_MAX_T 96         // waypoints the step kernel's register-resident phases are unrolled for (and its LDS holds)
#ifndef GTO_LIST_CAP
#define GTO_LIST_CAP 64      // wrench-list entries (8 doubles) per wave: a full chunk (64) fits after a drain
#endif

struct InstState {
  double f, lambda, nu, pred;
  double fgoal_try, fvel_try;
  int32_t slot, first, done, status, evals, argmin_try, argmin_cur, pad;
};

struct SolveParams {
  int32_t T, ts, use_standoff, n_max, max_iter, grad_mode;
  int32_t interleave;  // obstacle kernel: waypoints of a group nG apart (1) instead of consecutive (0); same results
  int32_t dbg_cut;  // debug: leave the obstacle kernel after phase k (1 prologue, 2 broad phase, 3 loop); 0 = off
  double dt, alpha, w_obstacle, w_vel, tol_step, tol_rel_f, lambda0;
};

struct BatchPtrs {
  // inputs (device)
  const int32_t* scene_id;  // [B]
  const double* qc;         // [B][ndof]
  const double* goals;      // [B][n_max][16]
  const int32_t* n_goals;   // [B]
  const double* standoff;   // [B][16] or null
  const double* base_pos;   // [B][3]
  const double* Q0;         // [B][ndof][T]
  // workspace
  InstState* state;  // [B]
  double* Qcur;      // [B][n][T]
  double* Qtry;      // [B][n][T]
  double* blocks;    // [2][B][T][BLK_STRIDE]
  double* goalblk;   // [2][B][2][BLK_STRIDE]
  double* ss_fixed;  // [B][4]  sum c^2 of the two pinned waypoints; of the static links under c_all / c_obs
  int32_t* n_done;   // [1]     instances that have finished
  double* qf;        // [B][T][F] joint value of every frame of the trial trajectory (0 for fixed joints)
  // slots: the solve loop keeps at most `cap` instances in flight.  slot_inst[i] is the instance slot i works on
  // (-1: none left); the step kernel of a slot whose instance finishes puts the next instance that has not
  // started into it (*next = id of that instance, n_total ids in all).  The joint values of the slot's trial
  // trajectory live in qfs, indexed by slot, so that the obstacle kernel needs no instance id to start its
  // kinematics.  Null outside the solve loop: the kernels then index the batch directly.
  int32_t* slot_inst;  // [cap]
  int32_t* next;       // [1]
  double* qfs;         // [cap][T][F]
  int32_t cap, n_total;
  long long* dbg;    // optional: phase timestamps of instance 0's step kernel (GTO_DEBUG_TIMING)
  // progress word in pinned host memory (or null): the first workgroup of every step launch publishes
  // (call tag << 32 | instances finished so far); the host polls it instead of copying n_done back through the stream
  unsigned long long* progress;
  unsigned long long progress_tag;
  unsigned long long* work;  // [64] or null: surface points gathered (one voxel record or field value each), in 64 cells by blockIdx
};

// ------------------------------------------------------------------------------------------------
// floor((x - o) / res) clipped to [0, n-1], bit-identical to the reference's index
// (gto/gto_models.py:174-187): the true quotient is only formed when the fast product lands within
// 1e-9 of a voxel face (the two can differ by an ulp there and nowhere else).
__device__ inline int voxel_axis(double x, double o, double res, double rinv, int n) {
  double d = x - o;
  double u = d * rinv;
  double k = floor(u);
  double fr = u - k;
  if (fr < 1e-9 || fr > 1.0 - 1e-9) k = floor(d / res);
  double hi = (double)(n - 1);
  k = (k >= 0.0) ? k : 0.0;  // NaN -> 0 like fmax(.,0)
  k = (k > hi) ? hi : k;
  return (int)k;
}
// Hot-loop form: u = y*rinv + cadd with cadd = (base - o)*rinv folded per workgroup (one FMA instead
// of add/sub/mul); any evaluation order is fine away from voxel faces, and within 1e-9 of a face the
// reference's own order (y + base - o) / res decides, so the index stays bit-identical.
__device__ inline int voxel_axis_fast(double y, double cadd, double base, double o, double res, double rinv, int n) {
  const double u = fma(y, rinv, cadd);
  double k = floor(u);
  const double fr = u - k;
  if (fabs(fr - 0.5) > 0.5 - 1e-9) k = floor(((y + base) - o) / res);
  int ki = (int)k;  // v_cvt_i32_f64 saturates, NaN -> 0
  return min(max(ki, 0), n - 1);
}

// ---- cross-lane sums without LDS traffic (gfx950: v_permlane32_swap / v_permlane16_swap / DPP)
typedef unsigned gto_uint2 __attribute__((ext_vector_type(2)));
// lanes 0-31 <- a summed over the two wave halves, lanes 32-63 <- b summed over the two halves
__device__ inline double swap32_add(double a, double b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const gto_uint2 l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  const gto_uint2 h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  return __hiloint2double(h.x, l.x) + __hiloint2double(h.y, l.y);
}
// rows (16 lanes) 0,2 <- a.row0+a.row1, a.row2+a.row3 ; rows 1,3 <- b.row0+b.row1, b.row2+b.row3
__device__ inline double swap16_add(double a, double b) {
  const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
  const gto_uint2 l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  const gto_uint2 h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  return __hiloint2double(h.x, l.x) + __hiloint2double(h.y, l.y);
}
template <int CTRL>
__device__ inline double dpp_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return v + __hiloint2double(hi2, lo2);
}
// sum over each 16-lane row, result in every lane of the row
__device__ inline double row_sum16(double v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v;
}
// Transpose-reduce of four per-lane values over the 64 lanes of a wave:
// afterwards row 0 holds sum(a), row 1 sum(c), row 2 sum(b), row 3 sum(d) in all of its lanes.
__device__ inline double wave_sum4(double a, double b, double c, double d) {
  return row_sum16(swap16_add(swap32_add(a, b), swap32_add(c, d)));
}

// Sum / maximum over the 64 lanes of a wave, result in every lane: two lane-swap steps (halves, then
// odd/even rows) and four DPP steps inside the 16-lane rows; 18 instructions, no LDS crossbar
// (the __shfl_xor butterfly costs 40+ and goes through ds_bpermute).
__device__ inline double wave_sum(double v) {
  const double h = swap32_add(v, v);
  return row_sum16(swap16_add(h, h));
}
template <int CTRL>
__device__ inline double dpp_max(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return fmax(v, __hiloint2double(hi2, lo2));
}
__device__ inline double wave_max(double v) {
  {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const gto_uint2 l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const gto_uint2 h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = fmax(__hiloint2double(h.x, l.x), __hiloint2double(h.y, l.y));
  }
  {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const gto_uint2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const gto_uint2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = fmax(__hiloint2double(h.x, l.x), __hiloint2double(h.y, l.y));
  }
  v = dpp_max<0xB1>(v);
  v = dpp_max<0x4E>(v);
  v = dpp_max<0x141>(v);
  v = dpp_max<0x140>(v);
  return v;
}

template <int CTRL>
__device__ inline int dpp_min_i32(int v) {
  return min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, 0xf, 0xf, false));
}
// minimum over the 64 lanes of a wave, result in every lane
__device__ inline int wave_min_i32(int v) {
  const gto_uint2 a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  v = min((int)a.x, (int)a.y);
  const gto_uint2 b = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = min((int)b.x, (int)b.y);
  v = dpp_min_i32<0xB1>(v);
  v = dpp_min_i32<0x4E>(v);
  v = dpp_min_i32<0x141>(v);
  v = dpp_min_i32<0x140>(v);
  return v;
}

__device__ inline void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Forward kinematics of TWO configurations at once by one wavefront: lanes 0-31 work on s_q[0..],
// s_fr[0..], lanes 32-63 on s_q[GTO_MAX_DOF..], s_fr[GTO_MAX_FRAMES*12..] (one frame after the other along the tree).
__device__ __forceinline__ void fk_pair_wave(const RobotDev* rb, const double* s_q2, double* s_fr2, int lane) {
  const int half = lane >> 5, l = lane & 31;
  const double* s_q = s_q2 + half * GTO_MAX_DOF;
  double* s_fr = s_fr2 + half * GTO_MAX_FRAMES * 12;
  const int F = rb->n_frames;
  if (l < F) {
    const int jt = rb->joint_type[l];
    const double* O = rb->origin[l];
    double* Lo = s_fr + 12 * l;
    if (jt == GTO_JOINT_REVOLUTE) {
      const double th = s_q[rb->q_index[l]];
      const double sn = sin(th), cs = cos(th), c1 = 1.0 - cs;
      const double u0 = rb->axis_unit[l][0], u1 = rb->axis_unit[l][1], u2 = rb->axis_unit[l][2];
      const double R00 = cs + c1 * u0 * u0, R01 = c1 * u0 * u1 - sn * u2, R02 = c1 * u0 * u2 + sn * u1;
      const double R10 = c1 * u1 * u0 + sn * u2, R11 = cs + c1 * u1 * u1, R12 = c1 * u1 * u2 - sn * u0;
      const double R20 = c1 * u2 * u0 - sn * u1, R21 = c1 * u2 * u1 + sn * u0, R22 = cs + c1 * u2 * u2;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double o0 = O[4 * r], o1 = O[4 * r + 1], o2 = O[4 * r + 2];
        Lo[4 * r] = o0 * R00 + o1 * R10 + o2 * R20;
        Lo[4 * r + 1] = o0 * R01 + o1 * R11 + o2 * R21;
        Lo[4 * r + 2] = o0 * R02 + o1 * R12 + o2 * R22;
        Lo[4 * r + 3] = O[4 * r + 3];
      }
    } else if (jt == GTO_JOINT_PRISMATIC) {
      const double qi = s_q[rb->q_index[l]];
      const double t0 = qi * rb->axis_unit[l][0], t1 = qi * rb->axis_unit[l][1], t2 = qi * rb->axis_unit[l][2];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double o0 = O[4 * r], o1 = O[4 * r + 1], o2 = O[4 * r + 2];
        Lo[4 * r] = o0;
        Lo[4 * r + 1] = o1;
        Lo[4 * r + 2] = o2;
        Lo[4 * r + 3] = o0 * t0 + o1 * t1 + o2 * t2 + O[4 * r + 3];
      }
    } else {
      for (int k = 0; k < 12; ++k) Lo[k] = O[k];
    }
  }
  wave_sync();
  if (l < 3) {
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
    int have = -1;
    for (int i = 0; i < F; ++i) {
      const int p = rb->parent[i];
      if (p < 0) {
        const double* Lr = s_fr + 12 * i + 4 * l;
        p0 = Lr[0];
        p1 = Lr[1];
        p2 = Lr[2];
        p3 = Lr[3];
        have = i;
        continue;
      }
      if (p != have) {
        wave_sync();
        const double* P = s_fr + 12 * p + 4 * l;
        p0 = P[0];
        p1 = P[1];
        p2 = P[2];
        p3 = P[3];
      }
      const double* Lm = s_fr + 12 * i;
      const double t0 = p0 * Lm[0] + p1 * Lm[4] + p2 * Lm[8];
      const double t1 = p0 * Lm[1] + p1 * Lm[5] + p2 * Lm[9];
      const double t2 = p0 * Lm[2] + p1 * Lm[6] + p2 * Lm[10];
      const double t3 = p0 * Lm[3] + p1 * Lm[7] + p2 * Lm[11] + p3;
      __builtin_amdgcn_wave_barrier();
      double* Oo = s_fr + 12 * i + 4 * l;
      Oo[0] = t0;
      Oo[1] = t1;
      Oo[2] = t2;
      Oo[3] = t3;
      p0 = t0;
      p1 = t1;
      p2 = t2;
      p3 = t3;
      have = i;
    }
  }
  wave_sync();
}

// world screw of optimised joint j from the frame that carries it: (a ; o x a) revolute, (0 ; a) prismatic
__device__ inline void screw_of_frame(const RobotDev* rb, int i, const double* F, double* s) {
  const double* u = rb->axis_unit[i];
  double a[3], o[3];
  for (int r = 0; r < 3; ++r) {
    a[r] = F[4 * r] * u[0] + F[4 * r + 1] * u[1] + F[4 * r + 2] * u[2];
    o[r] = F[4 * r + 3];
  }
  if (rb->joint_type[i] == GTO_JOINT_PRISMATIC) {
    s[0] = s[1] = s[2] = 0.0;
    s[3] = a[0];
    s[4] = a[1];
    s[5] = a[2];
  } else {
    double oxa[3];
    cross3(o, a, oxa);
    s[0] = a[0];
    s[1] = a[1];
    s[2] = a[2];
    s[3] = oxa[0];
    s[4] = oxa[1];
    s[5] = oxa[2];
  }
}

// quad permutation (4 consecutive lanes) of a double through DPP: no LDS traffic
template <int P0, int P1, int P2, int P3>
__device__ inline double quad_perm(double v) {
  constexpr int ctrl = P0 | (P1 << 2) | (P2 << 4) | (P3 << 6);
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// doubles of RobotDev::fk_tab in use: per frame O, c0, c1, K; per link Vo; per optimised joint U (16 each);
// then link_frame [L], opt_frame [n], prismatic flag [n], parent [F] stored as doubles
__host__ __device__ inline int fk_tab_doubles(int F, int L, int n) { return 64 * F + 16 * L + 16 * n + L + 2 * n + F; }
// LDS scratch (doubles) of fk_mfma_tree next to the table: X ping-pong [2][F][16], a dummy store target [64],
// ancestors [2][GTO_MAX_FRAMES] int
__host__ __device__ inline int fk_scratch_doubles(int F, int ng = 1) { return ng * 32 * F + 64 + GTO_MAX_FRAMES; }

// Forward kinematics of ONE configuration by a 256-thread workgroup on the FP64 matrix cores: visual
// transforms of the collision links (gto/gto_models.py:92-100) and world screws of the optimised joints.
//
// The frame transforms (optas/models.py:826-868) are products of 4x4 homogeneous matrices, and
// v_mfma_f64_4x4x4f64 computes four independent 4x4x4 FP64 products per instruction (a "block" is 16 lanes;
// measured lane maps, tools/probes/mfma_f64_probe.hip: A[i][k] in lane 16k + 4 blk + i, B[k][j] in lane
// 16k + 4 blk + j, D[i][j] in lane 16i + 4 blk + j).  A wave alone on its SIMD issues in order at roughly
// ten cycles per instruction, and a dependent FP64 result costs 35-50 cycles
// (tools/probes/fp64_latency_probe.hip): what counts is the number of instructions on the longest
// dependent path, not flops.  Hence: block = frame, wave = group of four frames, and the chain over the
// kinematic tree is a parallel prefix (pointer jumping): round r replaces G_f by G_anc(f) G_f and anc(f) by
// anc(anc(f)); after ceil(log2(depth)) rounds every G_f is global.  A round is one LDS exchange, one MFMA
// and one barrier.  Everything is kept transposed (X = G^T, so that a D result has the B layout): local
// X_f = M_f^T O_f^T, round X_f <- X_f X_anc.
//   M = c0 + cos c1 + sin K entrywise with c0 = h + u u^T, c1 = delta - u u^T, K = [u]x (Rodrigues,
//   optas/spatialmath.py:90-100); prismatic: sin := q, cos := 1, K = axis in the translation column.
// Outputs: block = link, V^T = Vo^T X_frame(link); block = optimised joint, [a o]^T = U^T X_frame(joint)
// with U = [u;0 | e4], screw = (a ; o x a) or (0 ; a) for a prismatic joint.
//   s_tab  LDS copy of RobotDev::fk_tab       s_sc [F][2] sin, cos | q, 1 | 0, 1 per frame
//   s_X    [2][F][16] + dummy [64] scratch    s_anc [2][GTO_MAX_FRAMES] scratch
// Every thread of the workgroup must call it (it contains barriers); the results are visible after it.
__device__ __forceinline__ void fk_mfma_tree(const RobotDev* __restrict__ rb, const double* __restrict__ s_tab, int ng,
                                    const double* __restrict__ s_sc, double* __restrict__ s_X, int* __restrict__ s_anc,
                                    int tid, double* __restrict__ s_vis, double* __restrict__ s_screw,
                                    long long* dbgp = nullptr, int scr_np = GTO_NB) {
  // `ng` configurations are advanced together, stage by stage, so that they share the barriers:
  // s_sc [ng][F][2], s_X [ng][2][F][16] then a dummy store target [64], s_anc [2][GTO_MAX_FRAMES] (the
  // ancestors do not depend on the configuration), s_vis [ng][L][12], s_screw [ng][scr_np][6]
  const int F = rb->n_frames, L = rb->n_links, n = rb->n_opt;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the compiler must see it as wave-uniform
  const int ra = lane >> 4, rc = lane & 3, blk = (lane >> 2) & 3;
  const int e = 4 * ra + rc, et = 4 * rc + ra;
  const double ident = (ra == rc) ? 1.0 : 0.0;
  const int nGf = (F + 3) >> 2;
  double* s_dummy = s_X + ng * 32 * F;  // target of the stores of blocks that have nothing to say
  const double* tVo = s_tab + 64 * F;
  const double* tU = tVo + 16 * L;
  const double* tI = tU + 16 * n;  // link_frame [L], opt_frame [n], prismatic flag [n], parent [F] as doubles
  constexpr int KW = GTO_MAX_FRAMES / 16;  // groups per wave
  int areg[KW];                            // ancestor of this lane's frame, carried from round to round
  if (dbgp && tid == 0) dbgp[0] = clock64();
  // local transforms X_f = (O_f M_f)^T: A operand M^T (lane supplies M[l>>4][l&3]), B operand O^T
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    const int g = wave + 4 * k;
    if (g >= nGf) continue;  // wave-uniform
    const int f0 = 4 * g + blk, f = f0 < F ? f0 : F - 1;
    const double* kt = s_tab + 64 * f;
    const double c0 = kt[16 + e], c1 = kt[32 + e], K = kt[48 + e], Ot = kt[et];
    areg[k] = (int)tI[L + 2 * n + f];
    for (int kq = 0; kq < ng; ++kq) {
      const double sn = s_sc[2 * (kq * F + f)], cs = s_sc[2 * (kq * F + f) + 1];
      const double Me = fma(sn, K, fma(cs, c1, c0));
      const double X = __builtin_amdgcn_mfma_f64_4x4x4f64(Me, Ot, 0.0, 0, 0, 0);
      *(f0 < F ? s_X + kq * 32 * F + 16 * f + e : s_dummy + lane) = X;
    }
    s_anc[f] = areg[k];  // sixteen lanes, one value
  }
  __syncthreads();
  if (dbgp && tid == 0) dbgp[1] = clock64();
  int cur = 0;
  for (int rd = 0; rd < rb->fk_rounds; ++rd) {
    const int* Ac = s_anc + cur * GTO_MAX_FRAMES;
    int* Aw = s_anc + (1 - cur) * GTO_MAX_FRAMES;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int g = wave + 4 * k;
      if (g >= nGf) continue;
      const int f0 = 4 * g + blk, f = f0 < F ? f0 : F - 1;
      const int a = areg[k], ac = a >= 0 ? a : 0;
      const int a2x = Ac[ac];
      for (int kq = 0; kq < ng; ++kq) {
        const double* Xc = s_X + kq * 32 * F + cur * 16 * F;
        double* Xw = s_X + kq * 32 * F + (1 - cur) * 16 * F;
        const double Aop = Xc[16 * f + et];  // A[i][k] = X_f[i][k]
        const double Bx = Xc[16 * ac + e];   // B[k][j] = X_anc[k][j]
        const double X = __builtin_amdgcn_mfma_f64_4x4x4f64(Aop, a >= 0 ? Bx : ident, 0.0, 0, 0, 0);
        *(f0 < F ? Xw + 16 * f + e : s_dummy + lane) = X;
      }
      areg[k] = a >= 0 ? a2x : -1;
      Aw[f] = areg[k];
    }
    __syncthreads();
    cur = 1 - cur;
    if (dbgp && tid == 0) dbgp[2 + rd] = clock64();
  }
  // output groups: first the links (four per MFMA), then the optimised joints, dealt round-robin to the waves
  const int nGl = (L + 3) >> 2, nGj = (n + 3) >> 2;
  for (int og = wave; og < nGl + nGj; og += 4) {
    if (og < nGl) {
      // D lane l holds V^T[l>>4][l&3] = V[l&3][l>>4]
      const int l1 = 4 * og + blk, l = l1 < L ? l1 : L - 1;
      const double Vo = tVo[16 * l + e];
      const int fl = (int)tI[l];
      for (int kq = 0; kq < ng; ++kq) {
        const double* Xg = s_X + kq * 32 * F + cur * 16 * F;  // X_f = G_f^T, row-major
        const double V = __builtin_amdgcn_mfma_f64_4x4x4f64(Vo, Xg[16 * fl + e], 0.0, 0, 0, 0);
        *((l1 < L && rc < 3) ? s_vis + (kq * L + l) * 12 + 4 * rc + ra : s_dummy + lane) = V;
      }
    } else {
      // lanes 0-15 of a block row hold a = R u, lanes 16-31 o = frame origin
      const int j1 = 4 * (og - nGl) + blk, j = j1 < n ? j1 : n - 1;
      const bool prism = tI[L + n + j] != 0.0;
      const double U = tU[16 * j + e];
      const int fj = (int)tI[L + j];
      const bool w = j1 < n && ra == 0 && rc < 3;
      for (int kq = 0; kq < ng; ++kq) {
        const double* Xg = s_X + kq * 32 * F + cur * 16 * F;
        const double S = __builtin_amdgcn_mfma_f64_4x4x4f64(U, Xg[16 * fj + e], 0.0, 0, 0, 0);
        const double av = S, ov = __shfl(S, (lane + 16) & 63, 64);
        const double a1 = quad_perm<1, 2, 0, 3>(av), a2 = quad_perm<2, 0, 1, 3>(av);
        const double o1 = quad_perm<1, 2, 0, 3>(ov), o2 = quad_perm<2, 0, 1, 3>(ov);
        const double cr = o1 * a2 - o2 * a1;
        double* sv = s_screw + (kq * scr_np + j) * 6;
        *(w ? sv + rc : s_dummy + lane) = prism ? 0.0 : av;
        *(w ? sv + 3 + rc : s_dummy + lane) = prism ? av : cr;
      }
    }
  }
  if (dbgp && tid == 0) dbgp[7] = clock64();
}

// ------------------------------------------------------------------------------------------------
// One-time per scene (gto_set_scene): voxel records {dx, dy, dz, c}.  Differences are formed in FP64
// from the float32 field with clipped neighbours exactly as gto/sdf_callback.py:90-114 indexes them;
// the division by 2*res stays in the solve kernel so the arithmetic order matches the oracle.
__global__ void k_build_records(const float* __restrict__ c, VoxelRec* __restrict__ rec, int nx, int ny, int nz) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nvox = (long)nx * ny * nz;
  if (i >= nvox) return;
  const int iz = (int)(i % nz), iy = (int)((i / nz) % ny), ix = (int)(i / ((long)nz * ny));
  const int ixp = min(ix + 1, nx - 1), ixm = max(ix - 1, 0);
  const int iyp = min(iy + 1, ny - 1), iym = max(iy - 1, 0);
  const int izp = min(iz + 1, nz - 1), izm = max(iz - 1, 0);
  auto at = [&](int x, int y, int z) { return (double)c[(long)z + (long)nz * ((long)y + (long)ny * x)]; };
  VoxelRec r;
  r.dx = at(ixp, iy, iz) - at(ixm, iy, iz);
  r.dy = at(ix, iyp, iz) - at(ix, iym, iz);
  r.dz = at(ix, iy, izp) - at(ix, iy, izm);
  r.c = c[i];
  r.pad = 0.f;
  rec[i] = r;
}

#define GTO_DIST_CAP 48  // voxels; beyond this the distance saturates

// Broad-phase support (one-time per scene): Chebyshev distance to the nearest non-zero record by
// iterated 3x3x3 min-plus-one relaxation (exact for the L-infinity metric after GTO_DIST_CAP sweeps).
__global__ void k_dist_init(const VoxelRec* __restrict__ rec, uint8_t* __restrict__ d, long nvox) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvox) return;
  const VoxelRec r = rec[i];
  d[i] = (r.c != 0.f || r.dx != 0.0 || r.dy != 0.0 || r.dz != 0.0) ? 0 : GTO_DIST_CAP;
}
__global__ void k_dist_relax(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int nx, int ny, int nz) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nvox = (long)nx * ny * nz;
  if (i >= nvox) return;
  const int iz = (int)(i % nz), iy = (int)((i / nz) % ny), ix = (int)(i / ((long)nz * ny));
  int best = in[i];
  for (int dx = -1; dx <= 1; ++dx)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dz = -1; dz <= 1; ++dz) {
        const int x = ix + dx, y = iy + dy, z = iz + dz;
        if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) continue;
        const int v = in[(long)z + (long)nz * ((long)y + (long)ny * x)] + 1;
        best = v < best ? v : best;
      }
  out[i] = (uint8_t)(best > GTO_DIST_CAP ? GTO_DIST_CAP : best);
}

// The same field in three launches instead of GTO_DIST_CAP sweeps: the Chebyshev metric is separable,
//   d(p) = min_{dx,dy,dz} max(|dx|, |dy|, |dz|, d0(p + d)) = min_dx max(|dx|, min_dy max(|dy|, min_dz max(|dz|, d0))),
// so one pass per axis with out(p) = min_{|o| <= cap} max(|o|, in(p + o e_axis)) is exact (max distributes over min);
// the scan stops as soon as |o| reaches the best value found.  Threads run along z (the fastest axis), so the loads of a
// wave are contiguous for every offset of every axis.
__global__ void k_dist_axis(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int nx, int ny, int nz, int axis) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nvox = (long)nx * ny * nz;
  if (i >= nvox) return;
  const int iz = (int)(i % nz), iy = (int)((i / nz) % ny), ix = (int)(i / ((long)nz * ny));
  const int pos = axis == 0 ? ix : (axis == 1 ? iy : iz), len = axis == 0 ? nx : (axis == 1 ? ny : nz);
  const long stride = axis == 0 ? (long)nz * ny : (axis == 1 ? (long)nz : 1L);
  int best = in[i];
  for (int o = 1; o < best; ++o) {  // |o| >= best cannot improve: max(|o|, .) >= best
    if (pos - o >= 0) {
      const int v = in[i - o * stride];
      const int c = v > o ? v : o;
      best = c < best ? c : best;
    }
    if (pos + o < len) {
      const int v = in[i + o * stride];
      const int c = v > o ? v : o;
      best = c < best ? c : best;
    }
  }
  out[i] = (uint8_t)best;
}

// ------------------------------------------------------------------------------------------------
// Dominant kernel.  One workgroup per (instance, group of TG consecutive waypoints):
//   grid.x = 8 * ceil(B/8) * ceil(nT/TG) (+ B goal workgroups);  b == blockIdx (mod 8), so all waypoints
//   of one instance (hence all gathers into one scene's field) are issued from one XCD and share its L2.
//   prologue  configuration -> forward kinematics of the TG waypoints IN PARALLEL inside the workgroup
//             (local transforms, then pointer jumping over the kinematic tree) -> visual transforms of the
//             collision links and joint screws staged in LDS.  Grouping waypoints amortises the latency
//             of this serial-ish part over TG times more surface-point work.
//   broad     one thread per (waypoint, chunk): bounding sphere vs Chebyshev distance field -> compacted
//             list of chunks that can touch a non-zero voxel
//   main loop one link-uniform chunk of 64 Morton-sorted surface points per wave step (sparse wrench lists)
//   epilogue  per-link 6x6 wrench Grams -> J^T J (n x n), J^T r (n), sum c^2 per waypoint
#define GTO_GOAL_SCRATCH(NP) (2 * GTO_MAX_FRAMES * 12 + 2 * GTO_MAX_DOF + 48 + 2 * (NP) * 6)  // doubles per goal wavefront
struct ObsLds {  // dynamic LDS layout (offsets in doubles), computed identically on host and device
  int vis, screw, uni, gram, list, out, active, total_doubles;
  __host__ __device__ ObsLds(int TG, int F, int L, int cap_active, int NP = GTO_NB) {
    const int stride = NP * NP + NP + 8;
    int o = 0;
    vis = o;    o += TG * L * 12;
    screw = o;  o += TG * NP * 6;
    // One region, two tenants.  Prologue: operand table, sin/cos [TG][F][2] and scratch of fk_mfma_tree (in the
    // goal workgroups: their scratch).  After the kinematics: Gram accumulators, wrench lists, surviving chunks.
    uni = o;
    const int fk = fk_tab_doubles(F, L, NP) + TG * F * 2 + fk_scratch_doubles(F, TG);
    gram = o;   o += TG * L * GTO_GRAM;  // every (waypoint, link) is folded by exactly one wave
    // wrench lists in the loop; in the epilogue s_u [TG][L][NP][6] and behind it the output blocks
    const int lst = 4 * GTO_LIST_CAP * 8, epi = TG * L * NP * 6 + TG * stride;
    list = o;   o += lst > epi ? lst : epi;
    out = list + TG * L * NP * 6;
    active = o; o += cap_active;  // int2 per entry
    total_doubles = (o - uni > fk ? o : uni + fk);
    if (total_doubles < 4 * GTO_GOAL_SCRATCH(NP)) total_doubles = 4 * GTO_GOAL_SCRATCH(NP);  // the goal workgroups: four wavefronts
  }
};

struct InstState;
template <int NP>
__device__ __forceinline__ void trial_goal_terms_wave(const RobotDev* rb, const BatchPtrs& bp, const SolveParams& sp, int B,
                                             int b, int lane, int trial, InstState* st, double* s_q, double* s_fr,
                                             double* s_gaff, double* s_gscr);

#ifndef GTO_OBS_MIN_WAVES
#define GTO_OBS_MIN_WAVES 5  // waves per SIMD the register allocator must leave room for: five workgroups per CU (31 KB of LDS each at three waypoints per workgroup)
#endif
template <int NP>
__global__ __launch_bounds__(256, GTO_OBS_MIN_WAVES) void k_obstacle_gram(const RobotDev* __restrict__ rb, const double* __restrict__ px,
                                                       const double* __restrict__ py, const double* __restrict__ pz,
                                                       const Chunk* __restrict__ chunks, const SceneDev* __restrict__ scenes,
                                                       BatchPtrs bp, SolveParams sp, int B, int t_begin, int nT,
                                                       int fixed_mode, int n_regular, int TG, int cap_active) {
  extern __shared__ __attribute__((aligned(16))) double smem_obs[];
  __shared__ int s_wcount[4];
  __shared__ int s_nactive;
  __shared__ unsigned s_touched[GTO_MAX_TG];  // per waypoint of the group: links whose Gram got a contribution

  const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = rb->n_links, n = rb->n_opt, T = sp.T, F = rb->n_frames;
  typedef Blk<NP> BK;
  const ObsLds lay(TG, F, L, cap_active, NP);
  double* s_vis = smem_obs + lay.vis;
  double* s_screw = smem_obs + lay.screw;
  double* s_gram = smem_obs + lay.gram;
  double* s_out = smem_obs + lay.out;
  double* s_list = smem_obs + lay.list;
  double* s_ktab = smem_obs + lay.uni;          // prologue only
  double* s_sc = s_ktab + fk_tab_doubles(F, L, n);  // prologue only
  int2* s_active = reinterpret_cast<int2*>(smem_obs + lay.active);  // (link | waypoint << 16, start | count << 16)
  double* s_u = s_list;    // [TG][L][NP][6] in the epilogue

  // Extra workgroups (blockIdx >= n_regular), one per instance: goal-set terms and velocity term of the
  // trial trajectory.  The step kernel only needs them at its NEXT launch, so they ride in the shadow
  // of the obstacle evaluation instead of sitting on the serial path between two launches.
  const bool listed = bp.slot_inst != nullptr && !fixed_mode;  // solve loop: workgroups are laid out over the slots
  const int n_act = listed ? bp.cap : B;
  if (bid >= n_regular) {
    // four instances per workgroup, one per wavefront: a goal workgroup holds a whole workgroup's LDS and wave
    // slots for one long serial job, so packing four of them into one costs a quarter of the slot time
    const int gi_ = (bid - n_regular) * 4 + wave;
    if (gi_ >= n_act) return;
    const int bg = listed ? bp.slot_inst[gi_] : gi_;
    if (bg < 0) return;  // empty slot
    // a fresh instance evaluates its seed, whose goal terms k_lm_init already produced
    if (bp.state[bg].done || (listed && bp.state[bg].first)) return;
    double* s_fr2 = smem_obs + wave * GTO_GOAL_SCRATCH(NP);  // [2][GTO_MAX_FRAMES*12]
    double* s_q2 = s_fr2 + 2 * GTO_MAX_FRAMES * 12;      // [2][GTO_MAX_DOF]
    double* s_ga = s_q2 + 2 * GTO_MAX_DOF;               // [48]
    double* s_gs = s_ga + 48;                            // [2][NP*6]
    trial_goal_terms_wave<NP>(rb, bp, sp, B, bg, lane, 1 - bp.state[bg].slot, bp.state + bg, s_q2, s_fr2, s_ga, s_gs);
    return;
  }
  // blockIdx -> (instance, waypoint group), bijective, with b % 8 == blockIdx % 8
  const int nG = (nT + TG - 1) / TG;
  const int xcd = bid & 7, kb = bid >> 3;
  const int bi = (kb / nG) * 8 + xcd;
  const int grp_id = kb % nG;
  if (bi >= n_act) return;
  const int b = listed ? bp.slot_inst[bi] : bi;
  if (b < 0) return;  // empty slot
  const InstState* st = bp.state + b;
  // fixed mode evaluates four "virtual waypoints": 0,1 = the two pinned waypoints (all links, value only);
  // 2,3 = the links no optimised joint moves, under c_all and under c_obs (their sum of c^2 is the same
  // at every waypoint and every iteration, so the solve loop never touches those points again)
  const int t0v = t_begin + grp_id * TG;                 // first (virtual) waypoint of this group
  const bool static_only = fixed_mode && t0v >= 2;
  const int t0w = static_only ? 0 : t0v;                 // waypoint whose configuration is used
  // Waypoints of the group: consecutive, or (solve loop, sp.interleave) nG apart, so that the few waypoints next to the
  // obstacles, which are neighbours in time, land in different workgroups.  The results do not depend on the grouping.
  const bool inter = sp.interleave && !fixed_mode;
  const int wstep = inter ? nG : 1, w0 = inter ? t_begin + grp_id : t0w;
  const int ng = static_only ? 1 : (inter ? (nT - grp_id + nG - 1) / nG : min(TG, (fixed_mode ? 2 : t_begin + nT) - t0v));  // waypoints in this group
  auto wp = [&](int kq_) { return w0 + kq_ * wstep; };    // waypoint of the group's kq-th member
  const bool dbg_wg = bp.dbg && b == 0 && grp_id == nG - 1;
  if (dbg_wg && tid == 0) bp.dbg[10] = clock64();

  // ---- prologue.  Every global load it needs is issued here, before the first branch that depends on one
  // of them: ONE memory round trip for the instance state, the joint values of the
  // frames (bp.qf, written by the step kernel) and the operand table of fk_mfma_tree.
  if (sp.dbg_cut == 6) return;
  const int done = listed ? 0 : st->done, slot_cur = st->slot;  // a slot never holds a finished instance
  // joint values of the group's waypoints: by slot in the solve loop (no dependence on the instance id)
  const double* __restrict__ qfp = listed ? bp.qfs + ((size_t)bi * T + w0) * F : bp.qf + ((size_t)b * T + w0) * F;
  const int qstride = (wstep - 1) * F;  // extra offset per group member
  const double qfv = tid < ng * F ? qfp[tid + (inter ? (tid / F) * qstride : 0)] : 0.0;
  const int jtv = tid < ng * F ? rb->joint_type[tid % F] : GTO_JOINT_FIXED;
  const int nt = fk_tab_doubles(F, L, n);  // rb->fk_tab is packed for exactly this (F, L, n)
  double tabv[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) tabv[u] = tid + 256 * u < nt ? rb->fk_tab[tid + 256 * u] : 0.0;
  if (done) return;
#pragma unroll
  for (int u = 0; u < 6; ++u)
    if (tid + 256 * u < nt) s_ktab[tid + 256 * u] = tabv[u];
  for (int k = tid + 256 * 6; k < nt; k += 256) s_ktab[k] = rb->fk_tab[k];  // very large robots
  if (sp.dbg_cut == 7) return;
  // sin/cos of every (waypoint, joint), one lane each
  if (tid < ng * F) {
    double a = 0.0, c = 1.0;
    if (jtv == GTO_JOINT_REVOLUTE) sincos(qfv, &a, &c);
    else if (jtv == GTO_JOINT_PRISMATIC) a = qfv;
    s_sc[2 * tid] = a;
    s_sc[2 * tid + 1] = c;
  }
  for (int idx = tid + 256; idx < ng * F; idx += 256) {  // waypoint groups of very large robots
    const int jt = rb->joint_type[idx % F];
    const double qv = qfp[idx + (idx / F) * qstride];
    double a = 0.0, c = 1.0;
    if (jt == GTO_JOINT_REVOLUTE) sincos(qv, &a, &c);
    else if (jt == GTO_JOINT_PRISMATIC) a = qv;
    s_sc[2 * idx] = a;
    s_sc[2 * idx + 1] = c;
  }
  if (tid < GTO_MAX_TG) s_touched[tid] = 0u;
  if (tid == 0) s_nactive = 0;
  __syncthreads();
  if (dbg_wg && tid == 0) bp.dbg[16] = clock64();
  if (sp.dbg_cut == 8) return;
  {
    double* s_X = s_sc + 2 * ng * F;
    fk_mfma_tree(rb, s_ktab, ng, s_sc, s_X, reinterpret_cast<int*>(s_X + ng * 32 * F + 64), tid, s_vis, s_screw,
                 dbg_wg ? bp.dbg + 20 : nullptr, NP);
  }
  __syncthreads();
  // the kinematics scratch is dead: its region now holds the Gram accumulators (the barriers of the broad
  // phase separate this from their first use)
  for (int i = tid; i < ng * L * GTO_GRAM; i += 256) s_gram[i] = 0.0;
  if (dbg_wg && tid == 0) bp.dbg[11] = clock64();
  if (sp.dbg_cut == 1) return;

  const SceneDev sc = scenes[bp.scene_id[b]];
  const double bx = bp.base_pos[3 * b], by = bp.base_pos[3 * b + 1], bz = bp.base_pos[3 * b + 2];
  const double cx = (bx - sc.ox) * sc.rinv, cy = (by - sc.oy) * sc.rinv, cz = (bz - sc.oz) * sc.rinv;
  const bool need_grad = !fixed_mode && sp.grad_mode == GTO_GRAD_CENTRAL_DIFF;
  const int nz = sc.nz;

  // ---- broad phase: one thread per (waypoint, chunk) transforms the chunk's bounding-sphere centre and
  // looks up the Chebyshev distance to the nearest non-zero voxel; a chunk whose sphere (radius R voxels,
  // +2 for the floor of the centre and index rounding) cannot reach one contributes exact zeros and is
  // skipped.  Survivors keep (waypoint, link) order (ballot prefix): a wave still sees few key changes.
  auto use_all = [&](int kq_) { return static_only ? (t0v == 2) : (wp(kq_) < sp.ts); };
  const int C = rb->n_chunks;
  for (int base_c = 0; base_c < ng * C; base_c += 256) {
    const int gi = base_c + tid;
    bool keep = false;
    int2 desc2 = make_int2(0, 0);
    if (gi < ng * C) {
      const int kq = gi / C, ci = gi % C;
      const Chunk cc = chunks[ci];
      desc2 = make_int2(cc.link | (kq << 16), cc.start | (cc.count << 16));
      const bool is_static = cc.pad != 0;  // link not moved by any optimised joint
      const double* V = s_vis + (kq * L + cc.link) * 12;
      const double u0 = (V[0] * cc.cx + V[1] * cc.cy + V[2] * cc.cz + V[3] + bx - sc.ox) * sc.rinv;
      const double u1 = (V[4] * cc.cx + V[5] * cc.cy + V[6] * cc.cz + V[7] + by - sc.oy) * sc.rinv;
      const double u2 = (V[8] * cc.cx + V[9] * cc.cy + V[10] * cc.cz + V[11] + bz - sc.oz) * sc.rinv;
      const int R = (int)ceil(cc.r * sc.rinv) + 2;
      const int k0 = (int)floor(u0), k1 = (int)floor(u1), k2 = (int)floor(u2);
      keep = true;
      if ((is_static && !fixed_mode) || (!is_static && static_only)) {
        keep = false;  // static links are accounted once at init; the static-only pass ignores the rest
      } else
      // only spheres that lie inside the grid (no clipped indices) and are closer than the cap can be culled
      if (R < GTO_DIST_CAP && k0 - R >= 0 && k1 - R >= 0 && k2 - R >= 0 && k0 + R < sc.nx && k1 + R < sc.ny && k2 + R < sc.nz) {
        const uint8_t* __restrict__ dist = use_all(kq) ? sc.d_all : sc.d_obs;
        const int dd = (int)dist[k2 + nz * (k1 + sc.ny * k0)];
        keep = dd <= R;
      }
    }
    const unsigned long long bm = __ballot(keep);
    if (lane == 0) s_wcount[wave] = __popcll(bm);
    __syncthreads();
    int woff = s_nactive;
    for (int w = 0; w < wave; ++w) woff += s_wcount[w];
    if (keep) {
      const int pos = woff + __popcll(bm & ((1ull << lane) - 1ull));
      if (pos < cap_active) s_active[pos] = desc2;
    }
    __syncthreads();
    if (tid == 0) s_nactive = min(s_nactive + s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3], cap_active);
    __syncthreads();
  }
  const int NA = s_nactive;
  // Contiguous range of surviving chunks per wave, cut at key changes only: a (waypoint, link) key is then
  // folded by exactly ONE wave, so a single Gram copy needs neither atomics nor per-wave copies, the result
  // is bit-reproducible, and the 15 KB of LDS saved buy a fourth workgroup per CU.  (A link has a handful
  // of chunks, so the ranges stay balanced.)
  auto cut_at_key = [&](int p) {
    if (p <= 0) return 0;
    while (p < NA && s_active[p].x == s_active[p - 1].x) ++p;
    return p < NA ? p : NA;
  };
  const int c0 = cut_at_key((int)(((long)NA * wave) / 4)), c1 = wave == 3 ? NA : cut_at_key((int)(((long)NA * (wave + 1)) / 4));
  if (dbg_wg && tid == 0) bp.dbg[12] = clock64();
  if (sp.dbg_cut == 2) return;

  // Sparse Gram accumulation.  Most surface points are in free space (zero gradient): a lane whose
  // point has a non-zero gradient appends x = (y x w, w, c) to a small per-wave LDS list, and the list is
  // folded into the per-link Gram X^T X (6x6 wrench Gram, c * wrench, c^2: the upper triangle of a 7x7) on
  // the FP64 matrix core: v_mfma_f64_16x16x4_f64 takes four list entries per instruction, with the same
  // register as A (A[i][k] = x_k[i], lane l: i = l & 15, k = l >> 4) and as B (B[k][j] = x_k[j]); the
  // accumulator D[row = (l >> 4) + 4 reg][col = l & 15] stays in registers until the (waypoint, link) key
  // changes.  The hot loop carries no per-lane accumulators, needs no cross-lane reduction, its cost
  // follows the number of points that actually touch the obstacle band, and the fold leaves the vector
  // ALU to the other waves of the CU.
  typedef double gto_v4f64 __attribute__((ext_vector_type(4)));
  double* lst = s_list + wave * (GTO_LIST_CAP * 8);
  double* gram_w = s_gram;
  const int mcol = lane & 15, mrow = lane >> 4;                // D column; D rows mrow (reg 0) and mrow + 4 (reg 1)
  // packed Gram index of D entry (row, col), row <= col < 7: 21 wrench-Gram entries, then c * wrench (6), then c^2
  // (the c^2 corner of the fold is not kept: slot 27 of a key holds the sum of c^2 over ALL its points, below)
  auto gram_index = [](int row, int col) { return col < 6 ? sym6(row, col) : (row < 6 ? 21 + row : -1); };
  const int gk0 = (mrow <= mcol && mcol < 7) ? gram_index(mrow, mcol) : -1;
  const int gk1 = (mrow + 4 <= mcol && mcol < 7) ? gram_index(mrow + 4, mcol) : -1;
  gto_v4f64 gD = {0.0, 0.0, 0.0, 0.0};
  double ss = 0.0;  // sum of c^2 over this lane's points of the current waypoint
  int cnt = 0, cur_key = -1;
  unsigned n_gathered = 0;  // surface points this wave looked up (wave-uniform)

#define GTO_DRAIN()                                                                          \
  do {                                                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                   \
    __builtin_amdgcn_wave_barrier();                                                         \
    for (int e_ = 0; e_ < cnt; e_ += 4) {                                                    \
      const int ei_ = e_ + mrow;                                                             \
      const double xv_ = (ei_ < cnt && mcol < 7) ? lst[ei_ * 8 + mcol] : 0.0;                \
      gD = __builtin_amdgcn_mfma_f64_16x16x4f64(xv_, xv_, gD, 0, 0, 0);                      \
    }                                                                                        \
    __builtin_amdgcn_wave_barrier();                                                         \
    cnt = 0;                                                                                 \
  } while (0)
  // key = link | waypoint << 16
#define GTO_FLUSH(key)                                                                       \
  do {                                                                                       \
    if (cnt) GTO_DRAIN();                                                                    \
    const int fk_ = (key) >> 16, fl_ = (key)&0xffff;                                         \
    double* gdst_ = gram_w + (fk_ * L + fl_) * GTO_GRAM;                                     \
    const bool w0_ = gk0 >= 0 && gD[0] != 0.0, w1_ = gk1 >= 0 && gD[1] != 0.0;              \
    if (w0_) gdst_[gk0] += gD[0];                                                            \
    if (w1_) gdst_[gk1] += gD[1];                                                            \
    if (__ballot(w0_ || w1_) && lane == 0) atomicOr(&s_touched[fk_], 1u << fl_);             \
    gD = gto_v4f64{0.0, 0.0, 0.0, 0.0};                                                      \
    /* sum of c^2 of the key: reduced per key, summed over the links in link order in the epilogue, so the  \
       value does not depend on how the chunks were dealt to the waves (nor on the group size) */ \
    const double sw_ = wave_sum(ss);                                                         \
    if (lane == 0) gdst_[27] += sw_;                                                         \
    ss = 0.0;                                                                                \
  } while (0)
  // Two-stage software pipeline over the wave's chunks.  Stage A of chunk c+1 (transform, voxel index,
  // ISSUE of the 32-B record gather) runs before stage B of chunk c (consume the record, append wrenches),
  // and the point coordinates of chunk c+2 are requested before that: three memory round trips (points,
  // records, points) are in flight at once instead of one after the other.
  struct ChunkLite {
    int key, start, count;
  };
  struct Staged {  // what stage B needs of a chunk
    int key, count;
    double y0, y1, y2;
    double4 rec;   // voxel record (gradient mode)
    float fval;    // field value (value-only mode)
  };
  auto load_chunk = [&](int c, ChunkLite& ch, double& x0, double& x1, double& x2) {
    const int2 d2 = s_active[c];
    ch = {d2.x, d2.y & 0xffff, d2.y >> 16};
    x0 = x1 = x2 = 0.0;
    if (lane < ch.count) {
      x0 = px[ch.start + lane];
      x1 = py[ch.start + lane];
      x2 = pz[ch.start + lane];
    }
  };
  auto stage_a = [&](const ChunkLite& ch, double x0, double x1, double x2, Staged& st_) {
    const int kq = ch.key >> 16, link = ch.key & 0xffff;
    const bool pre = use_all(kq);  // gto/gto_planner.py:117-131: c_all before the standoff waypoint
    const double* V = s_vis + (kq * L + link) * 12;
    // point in the robot-base frame (gto/gto_planner.py:114-116); the field frame adds base_position;
    // lanes past the end of the chunk carry x = 0 and have their cost and gradient zeroed in stage B
    const double y0 = V[0] * x0 + V[1] * x1 + V[2] * x2 + V[3];
    const double y1 = V[4] * x0 + V[5] * x1 + V[6] * x2 + V[7];
    const double y2 = V[8] * x0 + V[9] * x1 + V[10] * x2 + V[11];
    // voxel index (voxel_axis_fast for the three axes with ONE shared exact-fallback branch: if any axis
    // lands within 1e-9 of a voxel face, all three are redone in the reference's own order)
    const double u0 = fma(y0, sc.rinv, cx), u1 = fma(y1, sc.rinv, cy), u2 = fma(y2, sc.rinv, cz);
    double k0 = floor(u0), k1 = floor(u1), k2 = floor(u2);
    const double edge = fmax(fmax(fabs((u0 - k0) - 0.5), fabs((u1 - k1) - 0.5)), fabs((u2 - k2) - 0.5));
    if (edge > 0.5 - 1e-9) {
      k0 = floor(((y0 + bx) - sc.ox) / sc.res);
      k1 = floor(((y1 + by) - sc.oy) / sc.res);
      k2 = floor(((y2 + bz) - sc.oz) / sc.res);
    }
    const int ix = min(max((int)k0, 0), sc.nx - 1);  // v_cvt_i32_f64 saturates, NaN -> 0
    const int iy = min(max((int)k1, 0), sc.ny - 1);
    const int iz = min(max((int)k2, 0), sc.nz - 1);
    const int off = iz + nz * (iy + sc.ny * ix);
    st_.key = ch.key;
    st_.count = ch.count;
    st_.y0 = y0, st_.y1 = y1, st_.y2 = y2;
    st_.fval = 0.f;
    st_.rec = make_double4(0.0, 0.0, 0.0, 0.0);
    if (!need_grad) {
      st_.fval = (pre ? sc.c_all : sc.c_obs)[off];
    } else {
      // one 32-B voxel record: cost + central differences (gto/sdf_callback.py:90-114); the divisor
      // stays 2*res also at clipped borders.  Two 16-B loads, one cache line.
      st_.rec = *reinterpret_cast<const double4*>(&(pre ? sc.r_all : sc.r_obs)[off]);
    }
  };
  ChunkLite nch = {0, 0, 0};
  double n0 = 0.0, n1 = 0.0, n2 = 0.0;
  Staged cur = {};
  if (c0 < c1) {
    ChunkLite ch;
    double x0, x1, x2;
    load_chunk(c0, ch, x0, x1, x2);
    if (c0 + 1 < c1) load_chunk(c0 + 1, nch, n0, n1, n2);
    stage_a(ch, x0, x1, x2, cur);
  }
#pragma unroll 1
  for (int c = c0; c < c1; ++c) {
    Staged nxt = {};
    if (c + 1 < c1) {
      stage_a(nch, n0, n1, n2, nxt);  // gather of chunk c+1 goes out now
      if (c + 2 < c1) load_chunk(c + 2, nch, n0, n1, n2);
    }
    if (cur.key != cur_key) {
      if (cur_key >= 0) {
        GTO_FLUSH(cur_key);
      }
      cur_key = cur.key;
    }
    n_gathered += cur.count;
    {
      const bool valid = lane < cur.count;
      if (!need_grad) {
        const double cval = valid ? (double)cur.fval : 0.0;
        ss = fma(cval, cval, ss);
      } else {
        const double4 lo4 = cur.rec;
        const double y0 = cur.y0, y1 = cur.y1, y2 = cur.y2;
        const double cval = valid ? (double)__builtin_bit_cast(float, (unsigned)__double2loint(lo4.w)) : 0.0;
        ss = fma(cval, cval, ss);
        const double w0 = lo4.x * sc.inv2r;
        const double w1 = lo4.y * sc.inv2r;
        const double w2 = lo4.z * sc.inv2r;
        const bool act = valid && (w0 != 0.0 || w1 != 0.0 || w2 != 0.0);
        const unsigned long long am = __ballot(act);
        if (am) {  // wave-uniform
          if (act) {
            // wrench of the gradient about the base-frame origin: (y x w, w), then the cost value;
            // one 64-B list entry: three 16-B stores and one 8-B store
            double2* e = reinterpret_cast<double2*>(lst + (cnt + __popcll(am & ((1ull << lane) - 1ull))) * 8);
            e[0] = make_double2(y1 * w2 - y2 * w1, y2 * w0 - y0 * w2);
            e[1] = make_double2(y0 * w1 - y1 * w0, w0);
            e[2] = make_double2(w1, w2);
            reinterpret_cast<double*>(e)[6] = cval;
          }
          cnt += __popcll(am);
          if (cnt > GTO_LIST_CAP - 64) GTO_DRAIN();
        }
      }
    }
    cur = nxt;
  }
  if (cur_key >= 0) {
    GTO_FLUSH(cur_key);
  }
#undef GTO_FLUSH
#undef GTO_DRAIN
  // work counter of a profiled solve (bench.py prices the roofline on it): 64 cells so that the few thousand waves of a
  // launch that gathered anything do not queue up on one address
  if (bp.work && !fixed_mode && lane == 0 && n_gathered) atomicAdd(bp.work + (bid & 63), (unsigned long long)n_gathered);
  if (dbg_wg && tid == 0) bp.dbg[13] = clock64();
  if (sp.dbg_cut == 3) return;
  __syncthreads();
  // the wrench lists are dead: their region now holds s_u and, behind it, the output blocks
  for (int i = tid; i < ng * BK::STRIDE; i += 256) s_out[i] = 0.0;
  __syncthreads();
  // sum of c^2 per waypoint: its keys in link order
  if (tid < ng) {
    double v = 0.0;
    for (int l = 0; l < L; ++l) v += s_gram[(tid * L + l) * GTO_GRAM + 27];
    s_out[tid * BK::STRIDE + BK::SS] = v;
  }
  __syncthreads();

  // projection of the per-link wrench Grams onto the joint screws, over the links that were touched:
  //   JtJ[i][j] = sum_l [i,j in anc(l)] s_i^T W_l s_j ,  Jtr[i] = sum_l [i in anc(l)] s_i . v_l
  if (!fixed_mode) {
    // all waypoints of the group at once: s_u [ng][L][NP][6] in the dead list region
    for (int idx = tid; idx < ng * L * n; idx += 256) {
      const int kq = idx / (L * n), r_ = idx - kq * L * n, l = r_ / n, j = r_ - l * n;
      if (!((s_touched[kq] >> l) & 1u)) continue;
      const double* W = s_gram + (kq * L + l) * GTO_GRAM;
      const double* sj = s_screw + kq * NP * 6 + 6 * j;
      const bool on = (rb->link_anc[l] >> j) & 1u;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double u = 0.0;
        if (on) {
#pragma unroll
          for (int c = 0; c < 6; ++c) u += W[sym6(r, c)] * sj[c];
        }
        s_u[((kq * L + l) * NP + j) * 6 + r] = u;
      }
    }
    __syncthreads();
    // one thread per output entry, links summed in order (deterministic, no atomics)
    constexpr int NE = NP * NP + NP;  // entries of J^T J, then of J^T r
    for (int idx = tid; idx < ng * NE; idx += 256) {
      const int kq = idx / NE, e = idx - kq * NE;
      const unsigned touched = s_touched[kq];
      if (!touched) continue;  // the block stays zero
      const double* screw = s_screw + kq * NP * 6;
      if (e < NP * NP) {
        const int i = e / NP, j = e % NP;
        double v = 0.0;
        if (i < n && j < n) {
          const double* si = screw + 6 * i;
          for (int l = 0; l < L; ++l) {
            const uint32_t anc = rb->link_anc[l];
            if (((touched >> l) & 1u) && ((anc >> i) & 1u) && ((anc >> j) & 1u)) {
              const double* u = s_u + ((kq * L + l) * NP + j) * 6;
              v += si[0] * u[0] + si[1] * u[1] + si[2] * u[2] + si[3] * u[3] + si[4] * u[4] + si[5] * u[5];
            }
          }
        }
        s_out[kq * BK::STRIDE + BK::JTJ + e] = v;
      } else {
        const int i = e - NP * NP;
        double v = 0.0;
        if (i < n) {
          const double* si = screw + 6 * i;
          for (int l = 0; l < L; ++l)
            if (((touched >> l) & 1u) && ((rb->link_anc[l] >> i) & 1u)) {
              const double* vv = s_gram + (kq * L + l) * GTO_GRAM + 21;
              v += si[0] * vv[0] + si[1] * vv[1] + si[2] * vv[2] + si[3] * vv[3] + si[4] * vv[4] + si[5] * vv[5];
            }
        }
        s_out[kq * BK::STRIDE + BK::JTR + i] = v;
      }
    }
  }
  __syncthreads();
  if (dbg_wg && tid == 0) {
    bp.dbg[14] = clock64();
    bp.dbg[15] = NA;
  }
  if (fixed_mode) {
    if (tid < ng) bp.ss_fixed[4 * b + t0v + tid] = s_out[tid * BK::STRIDE + BK::SS];
  } else {
    // add the constant contribution of the static links (measured once at init)
    if (tid < ng) s_out[tid * BK::STRIDE + BK::SS] += bp.ss_fixed[4 * b + (wp(tid) < sp.ts ? 2 : 3)];
    __syncthreads();
    double* out = bp.blocks + (((size_t)(1 - slot_cur) * B + b) * T + w0) * BK::STRIDE;
    const int ostride = (wstep - 1) * BK::STRIDE;
    for (int i = tid; i < ng * BK::STRIDE; i += 256) out[i + (i / BK::STRIDE) * ostride] = s_out[i];
  }
}

// ------------------------------------------------------------------------------------------------
// Goal-set terms of one trajectory (gto/gto_planner.py:84-105), closed form in the moments of the
// gripper point cloud.  Executed by one wavefront.  gaff: [2][24] gripper+ee affines at the final
// and the standoff waypoint; scr: [2][n][6] screws at those waypoints.
struct GoalOut {
  double f_goal;
  int argmin;
};

__device__ __forceinline__ void goal_target(const double* grip_ee, const double* RT16, const double* S16, double* Y) {
  const double* Tg = grip_ee;
  const double* Te = grip_ee + 12;
  double inv[12], G[12], RT[12];
  for (int r = 0; r < 3; ++r) {  // invt (optas/spatialmath.py:271-280)
    for (int c = 0; c < 3; ++c) inv[4 * r + c] = Te[4 * c + r];
    inv[4 * r + 3] = -(Te[r] * Te[3] + Te[4 + r] * Te[7] + Te[8 + r] * Te[11]);
  }
  aff_mul(inv, Tg, G);
  for (int i = 0; i < 12; ++i) RT[i] = RT16[i];
  if (S16) {
    double S[12];
    for (int i = 0; i < 12; ++i) S[i] = S16[i];
    aff_mul(RT, S, RT);
  }
  aff_mul(RT, G, Y);
}

// sum_k || A p_k - Y p_k ||^2 = tr(D M D^T) + 2 d.(D mu) + K |d|^2, D = R_A - R_Y, d = t_A - t_Y
__device__ __forceinline__ double goal_cost_moments(const RobotDev* rb, const double* A, const double* Y) {
  double D[9], d[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) D[3 * r + c] = A[4 * r + c] - Y[4 * r + c];
    d[r] = A[4 * r + 3] - Y[4 * r + 3];
  }
  double s = 0.0;
  for (int r = 0; r < 3; ++r) {
    double dm[3];
    for (int c = 0; c < 3; ++c)
      dm[c] = D[3 * r] * rb->grip_M[c] + D[3 * r + 1] * rb->grip_M[3 + c] + D[3 * r + 2] * rb->grip_M[6 + c];
    s += dm[0] * D[3 * r] + dm[1] * D[3 * r + 1] + dm[2] * D[3 * r + 2];
    double dmu = D[3 * r] * rb->grip_mu[0] + D[3 * r + 1] * rb->grip_mu[1] + D[3 * r + 2] * rb->grip_mu[2];
    s += 2.0 * d[r] * dmu + rb->grip_count * d[r] * d[r];
  }
  return s;
}

// 6x6 Gram sum X^T X (packed upper, 21) and gradient sum X^T r (6) of the point-matching residuals
__device__ __forceinline__ void goal_gram_moments(const RobotDev* rb, const double* A, const double* Y, double* W21, double* v6) {
  const double K = rb->grip_count;
  double R[9], t[3], D[9], d[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      R[3 * r + c] = A[4 * r + c];
      D[3 * r + c] = A[4 * r + c] - Y[4 * r + c];
    }
    t[r] = A[4 * r + 3];
    d[r] = A[4 * r + 3] - Y[4 * r + 3];
  }
  double Rmu[3], Dmu[3], RM[9], XX[9], N[9];
  for (int r = 0; r < 3; ++r) {
    Rmu[r] = R[3 * r] * rb->grip_mu[0] + R[3 * r + 1] * rb->grip_mu[1] + R[3 * r + 2] * rb->grip_mu[2];
    Dmu[r] = D[3 * r] * rb->grip_mu[0] + D[3 * r + 1] * rb->grip_mu[1] + D[3 * r + 2] * rb->grip_mu[2];
    for (int c = 0; c < 3; ++c)
      RM[3 * r + c] = R[3 * r] * rb->grip_M[c] + R[3 * r + 1] * rb->grip_M[3 + c] + R[3 * r + 2] * rb->grip_M[6 + c];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      // sum x x^T = R M R^T + (R mu) t^T + t (R mu)^T + K t t^T ;  N = R M D^T
      XX[3 * r + c] = RM[3 * r] * R[3 * c] + RM[3 * r + 1] * R[3 * c + 1] + RM[3 * r + 2] * R[3 * c + 2] +
                      Rmu[r] * t[c] + t[r] * Rmu[c] + K * t[r] * t[c];
      N[3 * r + c] = RM[3 * r] * D[3 * c] + RM[3 * r + 1] * D[3 * c + 1] + RM[3 * r + 2] * D[3 * c + 2];
    }
  const double trxx = XX[0] + XX[4] + XX[8];
  double sx[3] = {Rmu[0] + K * t[0], Rmu[1] + K * t[1], Rmu[2] + K * t[2]};
  double W[36];
  for (int i = 0; i < 36; ++i) W[i] = 0.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) W[6 * r + c] = ((r == c) ? trxx : 0.0) - XX[3 * r + c];
  // upper-right block [sum x]_x
  W[6 * 0 + 4] = -sx[2];
  W[6 * 0 + 5] = sx[1];
  W[6 * 1 + 3] = sx[2];
  W[6 * 1 + 5] = -sx[0];
  W[6 * 2 + 3] = -sx[1];
  W[6 * 2 + 4] = sx[0];
  W[6 * 3 + 3] = W[6 * 4 + 4] = W[6 * 5 + 5] = K;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) W21[sym6(i, j)] = W[6 * i + j];
  // sum x x r = eps(N) + (R mu) x d + t x (D mu) + K t x d ;  sum r = D mu + K d
  double c1[3], c2[3], c3[3];
  cross3(Rmu, d, c1);
  cross3(t, Dmu, c2);
  cross3(t, d, c3);
  v6[0] = (N[5] - N[7]) + c1[0] + c2[0] + K * c3[0];
  v6[1] = (N[6] - N[2]) + c1[1] + c2[1] + K * c3[1];
  v6[2] = (N[1] - N[3]) + c1[2] + c2[2] + K * c3[2];
  v6[3] = Dmu[0] + K * d[0];
  v6[4] = Dmu[1] + K * d[1];
  v6[5] = Dmu[2] + K * d[2];
}

// One wavefront.  s_gaff [2][24], s_gscr [2][NP*6] in LDS.  goalblk_out: [2][Blk<NP>::STRIDE] or null.
template <int NP = GTO_NB>
__device__ __forceinline__ GoalOut goal_terms_wave(const RobotDev* rb, const SolveParams& sp, const double* goals, int n_goals,
                                          const double* standoff, const double* s_gaff, const double* s_gscr,
                                          double* goalblk_out, int lane) {
  // cost of every goal in the set, lanes stride over goals
  double best = INFINITY;
  int besti = 0x7fffffff;
  for (int g = lane; g < n_goals; g += 64) {
    double Y[12];
    goal_target(s_gaff, goals + 16 * g, nullptr, Y);
    double c = goal_cost_moments(rb, s_gaff, Y);
    if (sp.use_standoff) {
      goal_target(s_gaff + 24, goals + 16 * g, standoff, Y);
      c += goal_cost_moments(rb, s_gaff + 24, Y);
    }
    if (c < best) {
      best = c;
      besti = g;
    }
  }
  // first minimum over the wave (optas.mmin, gto/gto_planner.py:105)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ob = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(besti, o, 64);
    if (ob < best || (ob == best && oi < besti)) {
      best = ob;
      besti = oi;
    }
  }
  if (besti == 0x7fffffff) besti = 0;  // an empty or all-NaN goal set (the device entry point does not check its inputs)
  GoalOut out;
  out.f_goal = best;
  out.argmin = besti;
  if (goalblk_out) {
    // Gauss-Newton blocks of the arg-min goal: lanes 0-31 the final waypoint, lanes 32-63 the standoff
    const int n = rb->n_opt;
    const uint32_t anc = rb->frame_anc[rb->frame_gripper];
    const int which = lane >> 5, l = lane & 31;
    typedef Blk<NP> BK;
    double* blk = goalblk_out + which * BK::STRIDE;
    if (which == 1 && !sp.use_standoff) {
      for (int i = l; i < BK::STRIDE; i += 32) blk[i] = 0.0;
    } else {
      double Y[12], W21[21], v6[6];
      goal_target(s_gaff + 24 * which, goals + 16 * besti, which ? standoff : nullptr, Y);
      goal_gram_moments(rb, s_gaff + 24 * which, Y, W21, v6);
      const double* S = s_gscr + which * NP * 6;
#pragma unroll
      for (int h2 = 0; h2 < NP * NP / 32; ++h2) {
        const int e = l + 32 * h2, i = e / NP, j = e % NP;
        double v = 0.0;
        if (i < n && j < n && ((anc >> i) & 1u) && ((anc >> j) & 1u)) {
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            double u = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) u += W21[sym6(r, c)] * S[6 * j + c];
            v += S[6 * i + r] * u;
          }
        }
        blk[BK::JTJ + e] = v;
      }
      if (l < NP) {
        double g = 0.0;
        if (l < n && ((anc >> l) & 1u)) {
#pragma unroll
          for (int r = 0; r < 6; ++r) g += S[6 * l + r] * v6[r];
        }
        blk[BK::JTR + l] = g;
      }
    }
  }
  return out;
}

// Goal terms + velocity term of the trial trajectory; one wavefront per instance.  Forward kinematics
// is needed at two waypoints only (final and standoff); the obstacle kernel does its own.
// s_q [GTO_MAX_DOF], s_fr [GTO_MAX_FRAMES*12], s_gaff [48], s_gscr [2*NP*6] are LDS scratch.
template <int NP>
__device__ __forceinline__ void trial_goal_terms_wave(const RobotDev* rb, const BatchPtrs& bp, const SolveParams& sp, int B,
                                             int b, int lane, int trial, InstState* st, double* s_q, double* s_fr,
                                             double* s_gaff, double* s_gscr) {
  const int T = sp.T, n = rb->n_opt, ndof = rb->ndof;
  const double* Q0b = bp.Q0 + (size_t)b * ndof * T;
  const double* Qt = bp.Qtry + (size_t)b * n * T;
  {
    // both waypoints at once: lanes 0-31 the final waypoint, lanes 32-63 the standoff waypoint
    const int which = lane >> 5, l = lane & 31;
    const int t = which == 0 ? T - 1 : sp.ts;
    if (l < ndof) s_q[which * GTO_MAX_DOF + l] = Q0b[(size_t)l * T + t];
    wave_sync();
    if (l < n) s_q[which * GTO_MAX_DOF + rb->opt_index[l]] = Qt[(size_t)l * T + t];
    wave_sync();
    fk_pair_wave(rb, s_q, s_fr, lane);
    const double* fr = s_fr + which * GTO_MAX_FRAMES * 12;
    if (l < 12) {
      s_gaff[24 * which + l] = fr[12 * rb->frame_gripper + l];
      s_gaff[24 * which + 12 + l] = fr[12 * rb->frame_ee + l];
    }
    if (l >= 12 && l < 12 + n) {  // screw of optimised joint j = l - 12
      const int j = l - 12;
      for (int i = 0; i < rb->n_frames; ++i)
        if (rb->opt_of_frame[i] == j) screw_of_frame(rb, i, fr + 12 * i, s_gscr + which * NP * 6 + 6 * j);
    }
    wave_sync();
  }
  double* gblk = bp.goalblk + ((size_t)trial * B + b) * 2 * Blk<NP>::STRIDE;
  GoalOut go = goal_terms_wave<NP>(rb, sp, bp.goals + (size_t)b * sp.n_max * 16, bp.n_goals[b],
                               bp.standoff ? bp.standoff + (size_t)b * 16 : nullptr, s_gaff, s_gscr, gblk, lane);
  // velocity term with eliminated velocities (gto/gto_planner.py:133-135)
  double fv = 0.0;
  for (int idx = lane; idx < n * (T - 2); idx += 64) {
    const int j = idx / (T - 2), t = 1 + idx % (T - 2);
    const double v = (Qt[(size_t)j * T + t + 1] - Qt[(size_t)j * T + t]) / sp.dt;
    fv += v * v;
  }
  fv = wave_sum(fv);
  if (lane == 0) {
    st->fgoal_try = go.f_goal;
    st->fvel_try = sp.w_vel * fv;
    st->argmin_try = go.argmin;
  }
}

// raw != 0: take Q0's optimised rows as they are (evaluation entry points); otherwise build the seed.
template <int NP>
__global__ __launch_bounds__(256) void k_lm_init(const RobotDev* __restrict__ rb, BatchPtrs bp, SolveParams sp, int B, int raw) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ double s_gaff[48];
  __shared__ double s_gscr[2 * NP * 6];
  __shared__ double s_q[2 * GTO_MAX_DOF];
  __shared__ double s_fr[2 * GTO_MAX_FRAMES * 12];
  const int T = sp.T, n = rb->n_opt;
  InstState* st = bp.state + b;
  if (tid == 0) {
    st->f = INFINITY;
    st->lambda = sp.lambda0;
    st->nu = 2.0;
    st->pred = 0.0;
    st->slot = 0;
    st->first = 1;
    st->done = 0;
    st->status = GTO_STATUS_MAX_ITER;
    st->evals = 0;
    st->argmin_cur = 0;
  }
  if (bp.slot_inst && tid == 0) {  // the first `cap` instances take the slots; the rest wait for one to free up
    if (b < bp.cap) bp.slot_inst[b] = b;
    if (b == 0) *bp.next = bp.cap;
  }
  // seed: optimised rows of Q0, first two waypoints pinned to qc, the rest clipped into the bounds
  const double* Q0b = bp.Q0 + (size_t)b * rb->ndof * T;
  double* Qt = bp.Qtry + (size_t)b * n * T;
  double* Qc = bp.Qcur + (size_t)b * n * T;
  for (int idx = tid; idx < n * T; idx += 256) {
    const int j = idx / T, t = idx % T;
    double v = Q0b[(size_t)rb->opt_index[j] * T + t];
    if (!raw) {
      if (t < 2) v = bp.qc[(size_t)b * rb->ndof + rb->opt_index[j]];
      else v = fmin(fmax(v, rb->lower[j]), rb->upper[j]);
    }
    Qt[idx] = v;
    Qc[idx] = v;
  }
  __syncthreads();
  {  // joint value of every frame at every waypoint (parameter joints never change afterwards)
    const int F = rb->n_frames;
    double* qf = bp.qf + (size_t)b * T * F;
    for (int idx = tid; idx < T * F; idx += 256) {
      const int t = idx / F, i = idx - t * F, dq = rb->q_index[i];
      double v = 0.0;
      if (dq >= 0) {
        const int j = rb->opt_of_dof[dq];
        v = j >= 0 ? Qt[(size_t)j * T + t] : Q0b[(size_t)dq * T + t];
      }
      qf[idx] = v;
      if (bp.slot_inst && b < bp.cap) bp.qfs[(size_t)b * T * F + idx] = v;  // slot b starts with instance b
    }
  }
  if (tid < 64) trial_goal_terms_wave<NP>(rb, bp, sp, B, b, tid, 1, st, s_q, s_fr, s_gaff, s_gscr);
}

// 1/x to full double precision: hardware seed + two Newton steps (no IEEE division sequence)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// Gauss-Jordan inverse (no pivoting: SPD) of an 8x8 block held one entry per lane, lane = 8 r + c; pivot
// row/column moved by cross-lane shuffles; returns 1 if a pivot is not positive
__device__ __forceinline__ int gj_invert8(double& S, int lane, int r, int c) {
  int bad = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double pjj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(S), j * 9),
                                        __builtin_amdgcn_readlane(__double2loint(S), j * 9));
    const double prj = __shfl(S, (lane & 56) | j, 64);
    const double pjc = __shfl(S, j * 8 + c, 64);
    if (!(pjj > 0.0)) bad = 1;
    const double piv = fast_rcp(pjj);
    const double tcol = -prj * piv;
    const double in_row = (c == j) ? piv : pjc * piv;           // pivot row
    const double off_row = (c == j) ? tcol : fma(tcol, pjc, S);  // other rows
    S = (r == j) ? in_row : off_row;  // selects, not branches
  }
  return bad;
}
// y[r] = sum_c Z[r][c] z[c] for lane (r,c); every lane of row r ends up with y[r]
__device__ __forceinline__ double matvec8(double Z, double zc) {
  double pr = Z * zc;
  pr = dpp_add<0xB1>(pr);   // quad_perm [1,0,3,2]: lane ^ 1
  pr = dpp_add<0x4E>(pr);   // quad_perm [2,3,0,1]: lane ^ 2
  pr = dpp_add<0x141>(pr);  // row_half_mirror: the other quad of the 8-lane row
  return pr;
}

// dynamic LDS layout of k_lm_step (doubles): Z [m][64] | y [m][8] | e [m][8] | bfull [m][8] |
// x [m][8] | Q [8][T] | gaff [16] | red [16] ; then int act [m][8] ; then the kinematics scratch of
// One workgroup of four wavefronts per instance.  Lane (r,c) = (lane>>3, lane&7) of a wave owns entry
// (r,c) of the 8x8 blocks; the data-parallel phases (assembly, projected step, predicted decrease) are
// spread over the four waves by waypoint, the serial block recursion runs on wave 0.
// NW wavefronts per workgroup: 4, or 8 for launches with few instances in flight (the assembly, the projected step and the
// predicted decrease are spread over the waves; the twisted factorisation uses two of them either way).  The results do
// not depend on NW: the only cross-wave sum, the predicted decrease, is formed per waypoint class (s mod 8) in ascending
// order and the eight classes are added in a fixed order.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_lm_step(const RobotDev* __restrict__ rb, BatchPtrs bp, SolveParams sp, int B) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (blockIdx.x == 0 && threadIdx.x == 0 && bp.progress) {  // lagged by design: what had finished when this launch started
    const int nd = __hip_atomic_load(bp.n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(bp.progress, bp.progress_tag | (unsigned long long)(unsigned)nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const bool listed = bp.slot_inst != nullptr;  // solve loop: one workgroup per slot
  const int slot_id = blockIdx.x;
  const int b = listed ? bp.slot_inst[slot_id] : slot_id;
  if (b < 0) return;  // empty slot
  __shared__ int s_nid;
  InstState* st = bp.state + b;
  if (st->done) return;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int T = sp.T, n = rb->n_opt, m = T - 2;
  double* s_Z = smem;
  double* s_y = s_Z + (size_t)m * 64;
  double* s_e = s_y + m * 8;
  double* s_b = s_e + m * 8;
  double* s_x = s_b + m * 8;
  double* s_Q = s_x + m * 8;
  double* s_gaff = s_Q + 8 * T;
  double* s_red = s_gaff + 16;  // [32] cross-wave scratch: [0] failure flag, [1..8] step maxima, [8] first dense block (int, early), [16..23] predicted decrease by class
  int* s_actm = (int*)(s_red + 32);                // [m] frozen-joint bit masks (room for [m][8])
  int* s_first_dense = (int*)(s_red + 8);  // first waypoint block with an off-diagonal entry

  const int r = lane >> 3, c = lane & 7;
  const int trial = 1 - st->slot;
  const long long t_dbg0 = bp.dbg ? clock64() : 0;
  // P4 writes joint (tid & 7) of some waypoints: its frame, looked up long before it is needed
  const int nF = rb->n_frames, my_frame = (tid & 7) < n ? rb->opt_frame[tid & 7] : 0;
  // ... and its limits: every (waypoint, joint) item of this thread has joint index tid & 7
  const double my_lo = (tid & 7) < n ? rb->lower[tid & 7] : 0.0, my_hi = (tid & 7) < n ? rb->upper[tid & 7] : 0.0;
  double* __restrict__ qfb = listed ? bp.qfs + (size_t)slot_id * T * nF : bp.qf + (size_t)b * T * nF;  // joint values of the trial, by frame

  // ---- P0: objective of the trial point (every wave computes it: cheaper than a broadcast)
  double fo = 0.0;
  {
    const double* blk = bp.blocks + ((size_t)trial * B + b) * T * BLK_STRIDE;
    for (int t = 2 + lane; t < T; t += 64) fo += blk[(size_t)t * BLK_STRIDE + BLK_SS];
    fo = wave_sum(fo);
    fo += bp.ss_fixed[4 * b] + bp.ss_fixed[4 * b + 1];
  }
  const double f_try = st->fgoal_try + sp.w_obstacle * fo + st->fvel_try;

  // ---- P1: accept / reject (all threads take the same branch: inputs are block-uniform)
  double f = st->f, lambda = st->lambda, nu = st->nu;
  int slot = st->slot, done = 0, status = st->status;
  const int k = st->evals;
  const int argmin_try = st->argmin_try, argmin_cur0 = st->argmin_cur;
  const double pred0 = st->pred;
  bool accept = false;
  if (st->first) {
    accept = true;
  } else if (f_try < f && pred0 > 0.0) {
    accept = true;
    const double df = f - f_try, rho = df / pred0;
    const double sg = 2.0 * rho - 1.0;
    double fac = 1.0 - sg * sg * sg;
    fac = fmax(fac, 1.0 / 3.0);
    lambda = fmax(lambda * fac, 1e-12);
    nu = 2.0;
    if (df <= sp.tol_rel_f * (1.0 + f_try)) {
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
  double* __restrict__ Qc = bp.Qcur + (size_t)b * n * T;
  double* __restrict__ Qt = bp.Qtry + (size_t)b * n * T;
  if (accept) {
    f = f_try;
    slot = trial;
  }
  const int argmin_cur = accept ? argmin_try : argmin_cur0;
  // the iterate that is current from here on: loaded first (loads return in order), copied below
  constexpr int NQ = (8 * GTO_MAX_T + NT - 1) / NT;
  double qv[NQ];
  {
    const double* __restrict__ src = accept ? Qt : Qc;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      const int idx = tid + NT * u;
      qv[u] = (idx < n * T) ? src[idx] : 0.0;
    }
  }
  // global loads of P2 (normal equations at the iterate that is current from here on), issued now so that
  // their latency overlaps the trajectory copy below; wave w assembles waypoints s = w, w+NW, ...
  const double* __restrict__ oblk = bp.blocks + ((size_t)slot * B + b) * T * BLK_STRIDE;
  const double* __restrict__ gblk = bp.goalblk + ((size_t)slot * B + b) * 2 * BLK_STRIDE;
  const double alpha = sp.alpha;
  const bool inb = (r < n) && (c < n);
  constexpr int KMAX = (GTO_MAX_T - 2 + NW - 1) / NW;
  constexpr int NU = (8 * GTO_MAX_T + NT - 1) / NT;  // (waypoint, joint) items per thread
  double av[KMAX];  // undamped obstacle J^T J entry (r,c) of this wave's waypoints
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) {
    const int s = wave + NW * kk;
    av[kk] = (inb && s < m) ? oblk[(size_t)(s + 2) * BLK_STRIDE + BLK_JTJ + lane] : 0.0;
  }
  double jv[NU];  // obstacle J^T r of this thread's (waypoint, joint) items
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int idx = tid + NT * u, i = idx & 7;
    jv[u] = (idx < m * 8 && i < n) ? oblk[(size_t)((idx >> 3) + 2) * BLK_STRIDE + BLK_JTR + i] : 0.0;
  }
  const double gA0 = inb ? gblk[BLK_JTJ + lane] : 0.0;
  const double gA1 = (inb && sp.use_standoff) ? gblk[BLK_STRIDE + BLK_JTJ + lane] : 0.0;
  double gjv = 0.0;  // goal J^T r of both goal waypoints (threads 0..15), parked in LDS in P2
  if (tid < 16) {
    const int w = tid >> 3, i = tid & 7;
    if (i < n && (w == 0 || sp.use_standoff)) gjv = gblk[w * BLK_STRIDE + BLK_JTR + i];
  }
  // current iterate into LDS (rows >= n are padding); on accept it is the trial
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    const int idx = tid + NT * u;
    if (idx < 8 * T) s_Q[idx] = qv[u];
    if (accept && idx < n * T) Qc[idx] = qv[u];
  }
  if (!done && k >= sp.max_iter) {
    status = GTO_STATUS_MAX_ITER;
    done = 1;
  }
  // every state field is written exactly once, by thread 0, on whichever path leaves the kernel
#define GTO_FINISH(STATUS)                        \
  do {                                            \
    if (tid == 0) {                               \
      st->f = f;                                  \
      st->lambda = lambda;                        \
      st->nu = nu;                                \
      st->slot = slot;                            \
      st->first = 0;                              \
      st->done = 1;                               \
      st->status = (STATUS);                      \
      st->argmin_cur = argmin_cur;                \
      atomicAdd(bp.n_done, 1);                    \
      if (listed) { /* the freed slot takes the next instance that has not started yet */ \
        const int nid = atomicAdd(bp.next, 1);    \
        s_nid = nid < bp.n_total ? nid : -1;      \
        bp.slot_inst[slot_id] = s_nid;            \
      }                                           \
    }                                             \
    if (listed) { /* ... and the joint values of its seed (k_lm_init left them in qf) */ \
      __syncthreads();                            \
      const int nid = s_nid;                      \
      if (nid >= 0)                               \
        for (int i_ = tid; i_ < sp.T * rb->n_frames; i_ += NT)      \
          bp.qfs[(size_t)slot_id * sp.T * rb->n_frames + i_] = bp.qf[(size_t)nid * sp.T * rb->n_frames + i_]; \
    }                                             \
    return;                                       \
  } while (0)
  __syncthreads();
  if (done) GTO_FINISH(status);

  if (bp.dbg && b == 0 && tid == 0) bp.dbg[1] = clock64();
  // ---- P2: normal equations at the current iterate (A = J^T J, b = J^T r; f = sum r^2); its global loads
  // were issued in P1, as soon as the slot was known
  if (tid < 16) s_gaff[tid] = gjv;
  if (tid == 0) *s_first_dense = m;
  __syncthreads();
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[28] = clock64();
  int actv[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) actv[u] = 1;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int idx = tid + NT * u;
    if (idx < m * 8) {
      const int sI = idx >> 3, i = idx & 7, t = sI + 2;
      double bv = 0.0;
      int act = 1;  // padded rows count as frozen
      if (i < n) {
        bv = sp.w_obstacle * jv[u];
        if (t == T - 1) bv += s_gaff[i];
        if (sp.use_standoff && t == sp.ts) bv += s_gaff[8 + i];
        const double qt = s_Q[i * T + t], qm = s_Q[i * T + t - 1];
        bv += alpha * (qt - qm);
        if (t < T - 1) bv -= alpha * (s_Q[i * T + t + 1] - qt);
        // active set: on a bound with the descent direction pointing outward
        act = (qt <= my_lo && bv > 0.0) || (qt >= my_hi && bv < 0.0);
      }
      s_b[idx] = bv;
      actv[u] = act;
    }
  }
  // frozen variables of a waypoint as a bit mask (bit i = joint i): eight lanes of a ballot per waypoint
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int idx = tid + NT * u;
    const unsigned long long bal = __ballot(idx < m * 8 && actv[u] != 0);
    if (idx < m * 8 && (idx & 7) == 0) s_actm[idx >> 3] = (int)((bal >> (lane & 56)) & 0xffull);
  }
  __syncthreads();
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[29] = clock64();
  // undamped entries stay in registers (av[kk], read again by P5: same wave, same waypoints), the damped /
  // frozen system goes to s_Z; remember which blocks are purely diagonal.  Everything that depends on the
  // lane only is hoisted; the two waypoints that carry goal terms are patched by the wave that owns them.
  const bool diagl = r == c;
  const double dadd = (inb && diagl) ? 2.0 * alpha : 0.0;  // velocity term of an interior waypoint
  const double idv = diagl ? 1.0 : 0.0, dmul = diagl ? 1.0 + lambda : 1.0;
  const int lane_bits = (1 << r) | (1 << c);
  constexpr unsigned long long kOffDiag = ~0x8040201008040201ull;  // lanes (r,c) with r != c
  {
    int first = m;  // first dense block seen by this wave
    const int s_goal = T - 3, s_stand = sp.use_standoff ? sp.ts - 2 : -1;
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) {
      const int s = wave + NW * kk;
      if (s < m) {  // wave-uniform
        double a = fma(sp.w_obstacle, av[kk], dadd);
        if (s == s_goal) {  // goal waypoint T-1: goal block; its velocity term is alpha, not 2 alpha
          a += gA0;
          if (inb && diagl) a -= alpha;
        }
        if (s == s_stand) a += gA1;  // standoff waypoint
        const bool frozen = (s_actm[s] & lane_bits) != 0;
        const double v = frozen ? idv : a * dmul;
        av[kk] = a;
        s_Z[(size_t)s * 64 + lane] = v;
        if ((__ballot(v != 0.0) & kOffDiag) && s < first) first = s;
      }
    }
    if (lane == 0 && first < m) atomicMin(s_first_dense, first);
  }
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[30] = clock64();
  for (int idx = tid; idx < m * 8; idx += NT) {
    const int sI = idx >> 3, i = idx & 7;
    const int a0 = (s_actm[sI] >> i) & 1;
    const int a1 = (sI < m - 1) ? (s_actm[sI + 1] >> i) & 1 : 1;
    s_e[idx] = (a0 || a1) ? 0.0 : -alpha;
    s_y[idx] = a0 ? 0.0 : -s_b[idx];  // right-hand side
  }
  __syncthreads();
  const int s_dense = *s_first_dense;

  if (bp.dbg && b == 0 && tid == 0) bp.dbg[2] = clock64();
  // ---- P3: block-tridiagonal solve by the inverse-based Schur recursion, run from BOTH ends (twisted
  // factorisation): wave 0 eliminates downwards from waypoint 0, wave 1 upwards from the last waypoint,
  // they meet at block `mid`, and the two back-substitutions run outwards from there, again in parallel.
  // No extra arithmetic is needed and the serial chain of dense 8x8 inversions is roughly halved.
  //   down: S_s = D_s - E_{s-1} Z_{s-1} E_{s-1}, Z_s = S_s^{-1}, y_s = Z_s (rhs_s - E_{s-1} y_{s-1})
  //   up:   S_s = D_s - E_s Z_{s+1} E_s,         Z_s = S_s^{-1}, y_s = Z_s (rhs_s - E_s y_{s+1})
  //   mid:  S = D - E_{mid-1} Z_{mid-1} E_{mid-1} - E_mid Z_{mid+1} E_mid, x_mid = S^{-1}(rhs - E y - E y)
  //   back: x_s = y_s - Z_s E_s x_{s+1} (s < mid),  x_s = y_s - Z_s E_{s-1} x_{s-1} (s > mid)
  // Leading diagonal stretch (free-space waypoints carry only the velocity term, so S stays diagonal
  // until the first dense block): a scalar recurrence on the diagonal lanes.  Dense blocks:
  // Gauss-Jordan without pivoting (SPD) with the pivot row/column moved by cross-lane shuffles and the
  // mat-vec reductions done in registers; no LDS round trip, no barrier.
  // Meeting block: a diagonal step of the downward sweep costs about a tenth of a dense block, and every
  // block of the upward sweep is dense (it starts at the goal waypoint); balance the two chains.
  int mid;
  {
    const int sd = s_dense < m ? s_dense : m - 1;
    const int m2 = (10 * (m - 1) + 9 * sd) / 20;  // sd + 10 (mid - sd) = 10 (m - 1 - mid)
    const int m1 = (10 * (m - 1)) / 11;           // mid = 10 (m - 1 - mid)
    mid = m2 >= sd ? m2 : (m1 < sd ? m1 : sd);
    mid = mid < 0 ? 0 : (mid > m - 1 ? m - 1 : mid);
  }
  auto gj_invert = [&](double& S) -> int { return gj_invert8(S, lane, r, c); };
  auto matvec = [&](double Z, double zc) -> double { return matvec8(Z, zc); };
  if (wave == 0) {
    int fail = 0;
    double zp = 0.0, yp = 0.0;
    const int nd = s_dense < mid ? s_dense : mid;  // diagonal stretch of the downward sweep
    if (r == c) {
      for (int s = 0; s < nd; ++s) {
        const double ep = (s > 0) ? s_e[(s - 1) * 8 + r] : 0.0;
        const double S = s_Z[(size_t)s * 64 + lane] - ep * ep * zp;
        if (!(S > 0.0)) fail = 1;
        const double Zr = fast_rcp(S);
        const double y = Zr * (s_y[s * 8 + r] - ep * yp);
        s_Z[(size_t)s * 64 + lane] = Zr;
        s_x[s * 8 + r] = y;
        zp = Zr;
        yp = y;
      }
    }
    wave_sync();
    if (bp.dbg && b == 0 && tid == 0) bp.dbg[3] = clock64();
    double Zprev = (r == c) ? zp : 0.0;
    double yprev_c = (nd > 0) ? s_x[(nd - 1) * 8 + c] : 0.0;  // y_{s-1}[c] for lane (r,c)
    for (int s = nd; s < mid; ++s) {
      double S = s_Z[(size_t)s * 64 + lane];
      double zc = s_y[s * 8 + c];
      if (s > 0) {
        const double er = s_e[(s - 1) * 8 + r], ec = s_e[(s - 1) * 8 + c];
        S -= er * ec * Zprev;
        zc -= ec * yprev_c;
      }
      fail |= gj_invert(S);
      s_Z[(size_t)s * 64 + lane] = S;  // Z_s
      Zprev = S;
      const double pr = matvec(S, zc);
      if (c == 0) s_x[s * 8 + r] = pr;
      yprev_c = __shfl(pr, c << 3, 64);  // transpose: lane (r,c) picks y_s[c] from row c
    }
    if (lane == 0) s_red[0] = __any(fail) ? 1.0 : 0.0;
  } else if (wave == 1) {
    int fail = 0;
    double Zprev = 0.0, yprev_c = 0.0;
    for (int s = m - 1; s > mid; --s) {
      double S = s_Z[(size_t)s * 64 + lane];
      double zc = s_y[s * 8 + c];
      if (s < m - 1) {
        const double er = s_e[s * 8 + r], ec = s_e[s * 8 + c];
        S -= er * ec * Zprev;
        zc -= ec * yprev_c;
      }
      fail |= gj_invert(S);
      s_Z[(size_t)s * 64 + lane] = S;
      Zprev = S;
      const double pr = matvec(S, zc);
      if (c == 0) s_x[s * 8 + r] = pr;
      yprev_c = __shfl(pr, c << 3, 64);
    }
    if (lane == 0) s_red[1] = __any(fail) ? 1.0 : 0.0;
  }
  __syncthreads();
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[4] = clock64();
  if (wave == 0) {  // the meeting block
    int fail = (s_red[0] != 0.0) || (mid < m - 1 && s_red[1] != 0.0);
    double S = s_Z[(size_t)mid * 64 + lane];
    double zc = s_y[mid * 8 + c];
    if (mid > 0) {
      const double er = s_e[(mid - 1) * 8 + r], ec = s_e[(mid - 1) * 8 + c];
      S -= er * ec * s_Z[(size_t)(mid - 1) * 64 + lane];
      zc -= ec * s_x[(mid - 1) * 8 + c];
    }
    if (mid < m - 1) {
      const double er = s_e[mid * 8 + r], ec = s_e[mid * 8 + c];
      S -= er * ec * s_Z[(size_t)(mid + 1) * 64 + lane];
      zc -= ec * s_x[(mid + 1) * 8 + c];
    }
    fail |= gj_invert(S);
    const double pr = matvec(S, zc);
    if (c == 0) s_x[mid * 8 + r] = pr;  // x_mid
    if (lane == 0) s_red[0] = __any(fail) ? 1.0 : 0.0;
  }
  __syncthreads();
  if (s_red[0] != 0.0) GTO_FINISH(GTO_STATUS_NUMERICAL);
  if (wave == 0) {  // outwards to waypoint 0
    const int nd = s_dense < mid ? s_dense : mid;
    double xr = s_x[mid * 8 + r], xc = s_x[mid * 8 + c];
    for (int s = mid - 1; s >= nd; --s) {  // dense blocks
      const double pr = matvec(s_Z[(size_t)s * 64 + lane], s_e[s * 8 + c] * xc);
      xr = s_x[s * 8 + r] - pr;
      if (c == 0) s_x[s * 8 + r] = xr;
      xc = __shfl(xr, c << 3, 64);
    }
    if (r == c) {  // diagonal stretch
      for (int s = nd - 1; s >= 0; --s) {
        xr = s_x[s * 8 + r] - s_Z[(size_t)s * 64 + lane] * (s_e[s * 8 + r] * xr);
        s_x[s * 8 + r] = xr;
      }
    }
  } else if (wave == 1) {  // outwards to the last waypoint
    double xc = s_x[mid * 8 + c];
    for (int s = mid + 1; s < m; ++s) {
      const double pr = matvec(s_Z[(size_t)s * 64 + lane], s_e[(s - 1) * 8 + c] * xc);
      const double xr = s_x[s * 8 + r] - pr;
      if (c == 0) s_x[s * 8 + r] = xr;
      xc = __shfl(xr, c << 3, 64);
    }
  }
  __syncthreads();

  if (bp.dbg && b == 0 && tid == 0) bp.dbg[5] = clock64();
  // ---- P4: projected trial point; the LDS copy of Q becomes the trial, s_x the projected step
  double maxstep = 0.0;
  for (int idx = tid; idx < m * 8; idx += NT) {
    const int sI = idx >> 3, i = idx & 7, t = sI + 2;
    double sv = 0.0;
    if (i < n) {
      const double q0 = s_Q[i * T + t];
      double v = q0 + s_x[idx];
      v = fmin(fmax(v, my_lo), my_hi);
      Qt[(size_t)i * T + t] = v;
      qfb[(size_t)t * nF + my_frame] = v;  // the obstacle kernel reads joint values by frame
      s_Q[i * T + t] = v;
      sv = v - q0;
    }
    s_x[idx] = sv;
    maxstep = fmax(maxstep, fabs(sv));
  }
  if (tid < n * 2) Qt[(size_t)(tid >> 1) * T + (tid & 1)] = s_Q[(tid >> 1) * T + (tid & 1)];
  maxstep = wave_max(maxstep);
  if (lane == 0) s_red[1 + wave] = maxstep;
  __syncthreads();
  maxstep = s_red[1];
#pragma unroll
  for (int w = 1; w < NW; ++w) maxstep = fmax(maxstep, s_red[1 + w]);
  if (maxstep < sp.tol_step) GTO_FINISH(GTO_STATUS_CONVERGED);

  if (bp.dbg && b == 0 && tid == 0) bp.dbg[6] = clock64();
  // ---- P5: predicted decrease of the undamped model: -(2 b.s + s^T A s), waypoints split over waves.
  // Lane (r,c) adds A[r][c] s_r s_c; the lanes of column 0 also add the terms that are linear in s_r
  // (2 b_r s_r and the coupling -2 alpha s_r s'_r with the next waypoint), folded in as a lane constant.
  {
    static_assert(NW == 4 || NW == 8, "the waypoint classes of the predicted decrease are laid out for 4 or 8 waves");
    const double c0 = (c == 0) ? 2.0 : 0.0;
    double part0 = 0.0, part1 = 0.0;  // class wave (and, with four waves, class wave + 4: the odd kk)
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) {
      const int s = wave + NW * kk;
      if (s < m) {  // wave-uniform
        const double sr = s_x[s * 8 + r], scv = s_x[s * 8 + c];
        const double xn = (s < m - 1) ? s_x[(s + 1) * 8 + r] : 0.0;
        const double lin = c0 * fma(-alpha, xn, s_b[s * 8 + r]);
        if (NW == 8 || (kk & 1) == 0) part0 = fma(sr, fma(av[kk], scv, lin), part0);
        else part1 = fma(sr, fma(av[kk], scv, lin), part1);
      }
    }
    part0 = wave_sum(part0);
    if (lane == 0) s_red[16 + wave] = part0;
    if (NW == 4) {
      part1 = wave_sum(part1);
      if (lane == 0) s_red[16 + wave + 4] = part1;
    }
  }
  __syncthreads();
  const double acc = ((s_red[16] + s_red[17]) + (s_red[18] + s_red[19])) + ((s_red[20] + s_red[21]) + (s_red[22] + s_red[23]));
  if (tid == 0) {
    st->f = f;
    st->lambda = lambda;
    st->nu = nu;
    st->pred = -acc;
    st->slot = slot;
    st->first = 0;
    st->status = status;
    st->evals = k + 1;
    st->argmin_cur = argmin_cur;
  }
#undef GTO_FINISH
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[7] = clock64();
  // The goal-set and velocity terms of the new trial are evaluated by the extra workgroups of the next
  // k_obstacle_gram launch (off the serial path).
  if (bp.dbg && b == 0 && tid == 0) {
    bp.dbg[8] = clock64();
    bp.dbg[9] = s_dense;
    bp.dbg[0] = t_dbg0;
  }
}

// ------------------------------------------------------------------------------------------------
// Inverse kinematics (gto/ik_solver.py:30-110): T = 1,
//   min_q sum_k ||T_g(q) p_k - RT G p_k||^2 + w_obstacle sum_pts c_obs[off(x(q))],  lo <= q <= hi.
// One workgroup per goal pose runs the WHOLE projected Levenberg-Marquardt loop (same rules as the
// trajectory solve, oracle: solve_ik_instance): the problem has n <= 8 unknowns, so nothing leaves the
// chip between iterations.  Per evaluation: FK on the matrix cores (fk_mfma_tree), pose term in closed
// form in the gripper-cloud moments (goal_terms_wave with a one-goal set), collision term by the four
// waves over the link-uniform chunks of surface points: per link the sum of the cost and of the gradient
// wrenches (y x w, w), folded per wave and then in wave order (bit-reproducible); the collision term is
// the PLAIN sum of the cost and enters the gradient only: b = J^T r + (w/2) sum_l sum_{j in anc(l)} s_j . v_l.
__host__ __device__ inline int ik_lds_doubles(int F, int L, int n) {
  return fk_tab_doubles(F, L, n) + 2 * F + fk_scratch_doubles(F) + L * 12 + GTO_NB * 6 + 24 + 2 * BLK_STRIDE + 4 * L * 8 +
         2 * 64 + 2 * 8 + 8 + 8 + 8 + GTO_MAX_DOF + 16;
}

#ifndef GTO_IK_MIN_WAVES
#define GTO_IK_MIN_WAVES 3  // measured: 1 wave/SIMD (254 VGPRs) 268 k IK/s with the collision term, 2 (248, no scratch) 387 k, 3 (168 VGPRs, 288 B scratch) 423 k
#endif
__global__ __launch_bounds__(256, GTO_IK_MIN_WAVES) void k_ik_solve(const RobotDev* __restrict__ rb, const double* __restrict__ px,
                                                  const double* __restrict__ py, const double* __restrict__ pz,
                                                  const Chunk* __restrict__ chunks, const SceneDev* __restrict__ scenes,
                                                  const int32_t* __restrict__ scene_id, const double* __restrict__ q0,
                                                  const double* __restrict__ goals, const double* __restrict__ base_pos,
                                                  SolveParams sp, int B, double* __restrict__ q_out,
                                                  double* __restrict__ cost_out, int32_t* __restrict__ iters_out,
                                                  int32_t* __restrict__ status_out) {
  extern __shared__ __attribute__((aligned(16))) double smem_ik[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int F = rb->n_frames, L = rb->n_links, n = rb->n_opt, ndof = rb->ndof;
  double* s_tab = smem_ik;
  double* s_sc = s_tab + fk_tab_doubles(F, L, n);
  double* s_X = s_sc + 2 * F;
  double* s_vis = s_X + fk_scratch_doubles(F);
  double* s_screw = s_vis + L * 12;
  double* s_gaff = s_screw + GTO_NB * 6;  // gripper and ee affines
  double* s_gblk = s_gaff + 24;                // pose-term blocks (BLK_JTJ, BLK_JTR); goal_terms_wave clears a second one
  double* s_acc = s_gblk + 2 * BLK_STRIDE;     // [4 waves][L][8]: wrench sum (6), cost sum, -
  double* s_A = s_acc + 4 * L * 8;             // [2][64]
  double* s_b = s_A + 2 * 64;                  // [2][8]
  double* s_x = s_b + 2 * 8;                   // current iterate (optimised joints)
  double* s_xt = s_x + 8;                      // trial
  double* s_step = s_xt + 8;                   // projected step
  double* s_qf = s_step + 8;                   // full configuration (parameter joints as given)
  double* s_red = s_qf + GTO_MAX_DOF;          // [16]
  const bool collide = scene_id != nullptr;
  {
    const int nt = fk_tab_doubles(F, L, n);
    for (int k = tid; k < nt; k += 256) s_tab[k] = rb->fk_tab[k];
    if (tid < ndof) s_qf[tid] = q0[(size_t)b * ndof + tid];
    if (tid < 8) {
      double v = 0.0;
      if (tid < n) v = fmin(fmax(q0[(size_t)b * ndof + rb->opt_index[tid]], rb->lower[tid]), rb->upper[tid]);
      s_xt[tid] = v;
      s_x[tid] = v;
    }
  }
  SceneDev sc = {};
  double bx = 0.0, by = 0.0, bz = 0.0, cx = 0.0, cy = 0.0, cz = 0.0;
  if (collide) {
    sc = scenes[scene_id[b]];
    bx = base_pos[3 * b], by = base_pos[3 * b + 1], bz = base_pos[3 * b + 2];
    cx = (bx - sc.ox) * sc.rinv, cy = (by - sc.oy) * sc.rinv, cz = (bz - sc.oz) * sc.rinv;
  }
  SolveParams sp1 = sp;
  sp1.use_standoff = 0;
  const int r = lane >> 3, c = lane & 7;
  double f = INFINITY, lambda = sp.lambda0, nu = 2.0, pred = 0.0;
  int first = 1, k = 0, status = GTO_STATUS_MAX_ITER, slot = 0;  // slot: which of s_A/s_b holds the current iterate
  __syncthreads();
  for (;; ++k) {
    // ---- evaluate the trial configuration
    if (tid < F) {
      const int jt = rb->joint_type[tid], dq = rb->q_index[tid];
      double a = 0.0, cs = 1.0;
      if (dq >= 0) {
        const int j = rb->opt_of_dof[dq];
        const double qv = j >= 0 ? s_xt[j] : s_qf[dq];
        if (jt == GTO_JOINT_REVOLUTE) sincos(qv, &a, &cs);
        else if (jt == GTO_JOINT_PRISMATIC) a = qv;
      }
      s_sc[2 * tid] = a;
      s_sc[2 * tid + 1] = cs;
    }
    for (int i = tid; i < 4 * L * 8; i += 256) s_acc[i] = 0.0;
    __syncthreads();
    fk_mfma_tree(rb, s_tab, 1, s_sc, s_X, reinterpret_cast<int*>(s_X + 32 * F + 64), tid, s_vis, s_screw);
    if (tid < 24) {  // gripper and ee frames from the transposed results X_f = G_f^T (row-major)
      const double* Xg = s_X + (rb->fk_rounds & 1) * 16 * F;
      const int fsel = tid < 12 ? rb->frame_gripper : rb->frame_ee, e = tid % 12;
      s_gaff[tid] = Xg[16 * fsel + 4 * (e & 3) + (e >> 2)];
    }
    __syncthreads();
    double f_pos = 0.0;
    if (wave == 0) {
      const GoalOut go = goal_terms_wave(rb, sp1, goals + (size_t)b * 16, 1, nullptr, s_gaff, s_screw, s_gblk, lane);
      f_pos = go.f_goal;
      if (lane == 0) s_red[0] = f_pos;
    }
    if (collide) {
      const int nC = rb->n_chunks;
      const int c0 = (int)(((long)nC * wave) / 4), c1 = (int)(((long)nC * (wave + 1)) / 4);
      const bool need_grad = sp.grad_mode == GTO_GRAD_CENTRAL_DIFF;
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0;
      int cur_link = -1;
      auto flush = [&]() {
        const double t0 = wave_sum(a0), t1 = wave_sum(a1), t2 = wave_sum(a2), t3 = wave_sum(a3), t4 = wave_sum(a4),
                     t5 = wave_sum(a5), t6 = wave_sum(a6);
        if (lane == 0) {
          double* dst = s_acc + (wave * L + cur_link) * 8;
          dst[0] += t0, dst[1] += t1, dst[2] += t2, dst[3] += t3, dst[4] += t4, dst[5] += t5, dst[6] += t6;
        }
        a0 = a1 = a2 = a3 = a4 = a5 = a6 = 0.0;
      };
      for (int ci = c0; ci < c1; ++ci) {
        const int link = chunks[ci].link, start = chunks[ci].start, count = chunks[ci].count;
        if (link != cur_link) {
          if (cur_link >= 0) flush();
          cur_link = link;
        }
        if (lane < count) {
          const double x0 = px[start + lane], x1 = py[start + lane], x2 = pz[start + lane];
          const double* V = s_vis + link * 12;
          const double y0 = V[0] * x0 + V[1] * x1 + V[2] * x2 + V[3];
          const double y1 = V[4] * x0 + V[5] * x1 + V[6] * x2 + V[7];
          const double y2 = V[8] * x0 + V[9] * x1 + V[10] * x2 + V[11];
          const int ix = voxel_axis_fast(y0, cx, bx, sc.ox, sc.res, sc.rinv, sc.nx);
          const int iy = voxel_axis_fast(y1, cy, by, sc.oy, sc.res, sc.rinv, sc.ny);
          const int iz = voxel_axis_fast(y2, cz, bz, sc.oz, sc.res, sc.rinv, sc.nz);
          const int off = iz + sc.nz * (iy + sc.ny * ix);
          const double4 lo4 = *reinterpret_cast<const double4*>(&sc.r_obs[off]);
          a6 += (double)__builtin_bit_cast(float, (unsigned)__double2loint(lo4.w));
          if (need_grad) {
            const double w0 = lo4.x * sc.inv2r, w1 = lo4.y * sc.inv2r, w2 = lo4.z * sc.inv2r;
            a0 += y1 * w2 - y2 * w1;
            a1 += y2 * w0 - y0 * w2;
            a2 += y0 * w1 - y1 * w0;
            a3 += w0;
            a4 += w1;
            a5 += w2;
          }
        }
      }
      if (cur_link >= 0) flush();
    }
    __syncthreads();
    // ---- fold: normal equations and objective of the trial into slot 1 - slot
    const int ts_ = first ? slot : 1 - slot;
    if (tid < 64) s_A[ts_ * 64 + tid] = s_gblk[BLK_JTJ + tid];
    if (tid >= 64 && tid < 72) {
      const int i = tid - 64;
      double g = 0.0;
      if (i < n && collide) {
        const double* si = s_screw + 6 * i;
        for (int l = 0; l < L; ++l) {
          if (!((rb->link_anc[l] >> i) & 1u)) continue;
          double v[6];
#pragma unroll
          for (int e = 0; e < 6; ++e) v[e] = ((s_acc[(0 * L + l) * 8 + e] + s_acc[(1 * L + l) * 8 + e]) + s_acc[(2 * L + l) * 8 + e]) + s_acc[(3 * L + l) * 8 + e];
          g += si[0] * v[0] + si[1] * v[1] + si[2] * v[2] + si[3] * v[3] + si[4] * v[4] + si[5] * v[5];
        }
      }
      s_b[ts_ * 8 + i] = (i < n) ? s_gblk[BLK_JTR + i] + 0.5 * sp.w_obstacle * g : 0.0;
    }
    if (tid == 72) {
      double csum = 0.0;
      if (collide)
        for (int l = 0; l < L; ++l)
          csum += ((s_acc[(0 * L + l) * 8 + 6] + s_acc[(1 * L + l) * 8 + 6]) + s_acc[(2 * L + l) * 8 + 6]) + s_acc[(3 * L + l) * 8 + 6];
      s_red[1] = sp.w_obstacle * csum;
    }
    __syncthreads();
    const double f_try = s_red[0] + s_red[1];
    // ---- accept / reject (block-uniform)
    int done = 0;
    if (first) {
      first = 0;
      f = f_try;
      if (tid < 8) s_x[tid] = s_xt[tid];
    } else if (f_try < f && pred > 0.0) {
      const double df = f - f_try, rho = df / pred;
      f = f_try;
      slot = 1 - slot;
      if (tid < 8) s_x[tid] = s_xt[tid];
      const double sg = 2.0 * rho - 1.0;
      double fac = 1.0 - sg * sg * sg;
      fac = fmax(fac, 1.0 / 3.0);
      lambda = fmax(lambda * fac, 1e-12);
      nu = 2.0;
      if (df <= sp.tol_rel_f * (1.0 + f)) {
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
    if (k >= sp.max_iter) {
      status = GTO_STATUS_MAX_ITER;
      break;
    }
    __syncthreads();  // s_x visible
    // ---- step at the current iterate (wave 0): active set, damped system, Gauss-Jordan, projected trial
    if (wave == 0) {
      const double* A = s_A + slot * 64;
      const double* bb = s_b + slot * 8;
      const bool inb = r < n && c < n;
      const double xr = s_x[r], xc = s_x[c], br = bb[r], bc = bb[c];
      const bool ar = r >= n || (xr <= rb->lower[r < n ? r : 0] && br > 0.0) || (xr >= rb->upper[r < n ? r : 0] && br < 0.0);
      const bool ac = c >= n || (xc <= rb->lower[c < n ? c : 0] && bc > 0.0) || (xc >= rb->upper[c < n ? c : 0] && bc < 0.0);
      const double a = inb ? A[lane] : 0.0;
      double S = a;
      if (ar || ac) S = (r == c) ? 1.0 : 0.0;
      else if (r == c) S *= (1.0 + lambda);
      const int bad = gj_invert8(S, lane, r, c);
      const double dr = matvec8(S, ac ? 0.0 : -bc);  // delta[r]
      double v = fmin(fmax(xr + dr, rb->lower[r < n ? r : 0]), rb->upper[r < n ? r : 0]);
      if (r >= n) v = 0.0;
      const double sr = v - xr;
      const double sc_ = __shfl(sr, c << 3, 64);  // step[c]
      double part = a * sr * sc_;
      if (c == 0) part += 2.0 * br * sr;
      part = wave_sum(part);
      double ms = (c == 0 && r < n) ? fabs(sr) : 0.0;
      ms = wave_max(ms);
      if (c == 0) s_xt[r] = v;
      if (lane == 0) {
        s_red[2] = -part;
        s_red[3] = ms;
        s_red[4] = __any(bad) ? 1.0 : 0.0;
      }
      if (lane == 0 && __any(bad)) s_red[4] = 1.0;
    }
    __syncthreads();
    if (s_red[4] != 0.0) {
      status = GTO_STATUS_NUMERICAL;
      break;
    }
    if (s_red[3] < sp.tol_step) {
      status = GTO_STATUS_CONVERGED;
      break;
    }
    pred = s_red[2];
  }
  __syncthreads();
  if (tid < ndof) {
    const int j = rb->opt_of_dof[tid];
    q_out[(size_t)b * ndof + tid] = j >= 0 ? s_x[j] : s_qf[tid];
  }
  if (tid == 0) {
    if (cost_out) cost_out[b] = f;
    if (iters_out) iters_out[b] = k;
    if (status_out) status_out[b] = status;
  }
}

// ------------------------------------------------------------------------------------------------
// Base placement of a mobile manipulator (gto/base_planner.py:35-134):
//   min  w |(x,y,theta)|^2 + sum_i sum_k | A(q_i) p_k - B(x,y,theta) RT_i G p_k |^2 ,
//   lo <= q_i <= hi,  -pi <= theta <= pi,   unknowns: the base pose and one arm configuration per goal.
// One workgroup per goal set runs the whole projected Levenberg-Marquardt loop (oracle:
// solve_base_instance).  Residuals r_k = x_k - tau_k with x_k = A(q_i) p_k (moved by the joint screws s_j)
// and tau_k = B RT_i G p_k (moved by the base pose through the screws sigma = (0;e_x), (0;e_y), (z; -z x t)),
// so with X_p = [-[p]_x | I] the Gauss-Newton blocks of goal i are
//   D = S^T (sum X_x^T X_x) S,  C = -S^T (sum X_x^T X_tau) Sigma,  S_b = Sigma^T (sum X_tau^T X_tau) Sigma,
// all closed forms in the moments of the gripper cloud (goal_gram_moments twice, goal_cross_moments).
// The normal equations are an arrow: one 8x8 block D_i per goal, coupled only through the 3x3 base
// block; each wave eliminates its goals' blocks by Gauss-Jordan, thread 0 solves the 3x3 Schur
// complement, and sums over goals run in goal order (bit-reproducible).
#define GTO_MAX_BASE_GOALS 32
#define GTO_BASE_SYS 112  // per goal: D 8x8 | C 8x3 | S 3x3 | g 3+8 | f
__host__ __device__ inline int base_lds_doubles(int n_max) {
  return GTO_MAX_DOF + 8 * GTO_MAX_DOF + 8 * GTO_MAX_FRAMES * 12 + 8 * 24 + 8 * GTO_NB * 6 + 2 * n_max * GTO_BASE_SYS +
         3 * (8 + 8 * n_max) + n_max * (24 + 8 + 16) + 32;
}

// sum_k X_x^T X_tau (6x6, row-major) for x = A p, tau = Y p:
//   [[ tr(T) I - T , [sum x]_x ], [ -[sum tau]_x , K I ]],  T = sum tau x^T = R_Y M R_A^T + (R_Y mu) t_A^T + t_Y (R_A mu)^T + K t_Y t_A^T
__device__ __forceinline__ void goal_cross_moments(const RobotDev* rb, const double* A, const double* Y, double* Wc) {
  const double K = rb->grip_count;
  double Amu[3], Ymu[3], YM[9], TX[9];
  for (int r = 0; r < 3; ++r) {
    Amu[r] = A[4 * r] * rb->grip_mu[0] + A[4 * r + 1] * rb->grip_mu[1] + A[4 * r + 2] * rb->grip_mu[2];
    Ymu[r] = Y[4 * r] * rb->grip_mu[0] + Y[4 * r + 1] * rb->grip_mu[1] + Y[4 * r + 2] * rb->grip_mu[2];
    for (int c = 0; c < 3; ++c)
      YM[3 * r + c] = Y[4 * r] * rb->grip_M[c] + Y[4 * r + 1] * rb->grip_M[3 + c] + Y[4 * r + 2] * rb->grip_M[6 + c];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      TX[3 * r + c] = YM[3 * r] * A[4 * c] + YM[3 * r + 1] * A[4 * c + 1] + YM[3 * r + 2] * A[4 * c + 2] + Ymu[r] * A[4 * c + 3] +
                      Y[4 * r + 3] * Amu[c] + K * Y[4 * r + 3] * A[4 * c + 3];
  const double tr = TX[0] + TX[4] + TX[8];
  const double sx[3] = {Amu[0] + K * A[3], Amu[1] + K * A[7], Amu[2] + K * A[11]};
  const double st[3] = {Ymu[0] + K * Y[3], Ymu[1] + K * Y[7], Ymu[2] + K * Y[11]};
  for (int i = 0; i < 36; ++i) Wc[i] = 0.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Wc[6 * r + c] = ((r == c) ? tr : 0.0) - TX[3 * r + c];
  Wc[6 * 0 + 4] = -sx[2], Wc[6 * 0 + 5] = sx[1], Wc[6 * 1 + 3] = sx[2], Wc[6 * 1 + 5] = -sx[0], Wc[6 * 2 + 3] = -sx[1], Wc[6 * 2 + 4] = sx[0];
  Wc[6 * 3 + 1] = st[2], Wc[6 * 3 + 2] = -st[1], Wc[6 * 4 + 0] = -st[2], Wc[6 * 4 + 2] = st[0], Wc[6 * 5 + 0] = st[1], Wc[6 * 5 + 1] = -st[0];
  Wc[6 * 3 + 3] = Wc[6 * 4 + 4] = Wc[6 * 5 + 5] = K;
}

// screw of variable a of one goal's block: a < 3 the base pose acting on the target points, a >= 3 joint a - 3
__device__ __forceinline__ void base_screw(int a, const double* scr, double tx, double ty, double* s) {
  if (a < 3) {
    s[0] = s[1] = 0.0, s[2] = a == 2 ? 1.0 : 0.0;
    s[3] = a == 0 ? 1.0 : (a == 2 ? ty : 0.0);
    s[4] = a == 1 ? 1.0 : (a == 2 ? -tx : 0.0);
    s[5] = 0.0;
  } else {
#pragma unroll
    for (int q = 0; q < 6; ++q) s[q] = scr[6 * (a - 3) + q];
  }
}

#ifndef GTO_BASE_MIN_WAVES
#define GTO_BASE_MIN_WAVES 2  // measured (tools/ab_base.sh, Fetch, 10 goals per set; 64 / 1024 sets per call, w = 0 | 0.01): 1 wave/SIMD (256 VGPRs + 36 AGPRs) 39 k / 364 k | 17.5 k / 79 k sets/s; 2 (256 VGPRs, 108 B scratch) 38 k / 411 k | 17.0 k / 126 k; 3 (168, 464 B) 31 k / 333 k | 14.8 k / 99 k
#endif
__global__ __launch_bounds__(256, GTO_BASE_MIN_WAVES) void k_base_solve(const RobotDev* __restrict__ rb, const double* __restrict__ qc,
                                                    const double* __restrict__ goals, const int32_t* __restrict__ n_goals,
                                                    SolveParams sp, double w_effort, int n_max, double* __restrict__ y_out,
                                                    double* __restrict__ q_out, double* __restrict__ cost_out,
                                                    int32_t* __restrict__ iters_out, int32_t* __restrict__ status_out,
                                                    const double* __restrict__ y0, const double* __restrict__ q0) {
  // y0 [B][3], q0 [B][n_max][ndof]: start point instead of the reference's (zero pose, qc for every goal); null in
  // the solve, set by gto_eval_base_objective (a run capped at 0 iterations returns the objective at its start)
  extern __shared__ __attribute__((aligned(16))) double smem_base[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int F = rb->n_frames, n = rb->n_opt, ndof = rb->ndof, ng = n_goals[b], NV = 8 + 8 * n_max;
  double* s_qc = smem_base;
  double* s_q = s_qc + GTO_MAX_DOF;             // [8][GTO_MAX_DOF]
  double* s_fr = s_q + 8 * GTO_MAX_DOF;         // [8][GTO_MAX_FRAMES*12]
  double* s_gaff = s_fr + 8 * GTO_MAX_FRAMES * 12;  // [8][24] gripper and ee affines
  double* s_scr = s_gaff + 8 * 24;              // [8][GTO_NB*6] joint screws
  double* s_sys = s_scr + 8 * GTO_NB * 6;  // [2][n_max][GTO_BASE_SYS]
  double* s_x = s_sys + 2 * n_max * GTO_BASE_SYS;  // [8 + 8 n_max]: base pose in 0..2, goal i's joints at 8 + 8 i
  double* s_xt = s_x + NV;
  double* s_st = s_xt + NV;                     // projected step
  double* s_E = s_st + NV;                      // [n_max][24] D^-1 C
  double* s_u = s_E + n_max * 24;               // [n_max][8]  D^-1 rhs
  double* s_part = s_u + n_max * 8;             // [n_max][16] C^T E (9), C^T u (3), quadratic part, max step
  double* s_red = s_part + n_max * 16;          // [32]
  const double PI_ = 3.141592653589793;
  if (tid < ndof) s_qc[tid] = qc[(size_t)b * ndof + tid];
  for (int i = tid; i < NV; i += 256) {
    double v = 0.0;
    if (i >= 8) {
      const int j = i & 7, gi = (i >> 3) - 1;
      const double* src = q0 ? q0 + ((size_t)b * n_max + gi) * ndof : qc + (size_t)b * ndof;
      if (j < n) v = fmin(fmax(src[rb->opt_index[j]], rb->lower[j]), rb->upper[j]);
    } else if (y0 && i < 3) {
      v = y0[3 * (size_t)b + i];
      if (i == 2) v = fmin(fmax(v, -PI_), PI_);
    }
    s_x[i] = v;
    s_xt[i] = v;
    s_st[i] = 0.0;
  }
  const int half = lane >> 5, l = lane & 31, slot8 = wave * 2 + half;
  const int r = lane >> 3, c = lane & 7;
  const uint32_t ancg = rb->frame_anc[rb->frame_gripper];
  double f = INFINITY, lambda = sp.lambda0, nu = 2.0, pred = 0.0;
  int first = 1, k = 0, status = GTO_STATUS_MAX_ITER, slot = 0;
  __syncthreads();
  for (;; ++k) {
    // ---- evaluate the trial point: eight goals per pass, two per wavefront
    const int ts_ = first ? slot : 1 - slot;
    const double th = s_xt[2], bxp = s_xt[0], byp = s_xt[1];
    double sn, cs;
    sincos(th, &sn, &cs);
    for (int pass = 0; pass * 8 < ng; ++pass) {
      const int gi = pass * 8 + slot8;
      const bool valid = gi < ng;
      if (l < ndof) s_q[slot8 * GTO_MAX_DOF + l] = s_qc[l];
      wave_sync();
      if (l < n && valid) s_q[slot8 * GTO_MAX_DOF + rb->opt_index[l]] = s_xt[8 + 8 * gi + l];
      wave_sync();
      fk_pair_wave(rb, s_q + wave * 2 * GTO_MAX_DOF, s_fr + wave * 2 * GTO_MAX_FRAMES * 12, lane);
      const double* fr = s_fr + slot8 * GTO_MAX_FRAMES * 12;
      if (l < 12) {
        s_gaff[24 * slot8 + l] = fr[12 * rb->frame_gripper + l];
        s_gaff[24 * slot8 + 12 + l] = fr[12 * rb->frame_ee + l];
      }
      if (l >= 12 && l < 12 + n) {
        const int j = l - 12;
        for (int i = 0; i < F; ++i)
          if (rb->opt_of_frame[i] == j) screw_of_frame(rb, i, fr + 12 * i, s_scr + slot8 * GTO_NB * 6 + 6 * j);
      }
      wave_sync();
      if (valid) {
        const double* ga = s_gaff + 24 * slot8;
        const double* scr = s_scr + slot8 * GTO_NB * 6;
        double Y0[12], Y[12];
        goal_target(ga, goals + ((size_t)b * n_max + gi) * 16, nullptr, Y0);
        // target pose seen from the current base: B RT G, B = rt2tr(rotz(theta), [x, y, 0])
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          Y[cc] = cs * Y0[cc] - sn * Y0[4 + cc] + (cc == 3 ? bxp : 0.0);
          Y[4 + cc] = sn * Y0[cc] + cs * Y0[4 + cc] + (cc == 3 ? byp : 0.0);
          Y[8 + cc] = Y0[8 + cc];
        }
        const double fi = goal_cost_moments(rb, ga, Y);
        double* sys = s_sys + ((size_t)ts_ * n_max + gi) * GTO_BASE_SYS;
        // entry e of the block: D (0..63), C (64..87), S_b (88..96), gradient half (97..107)
        auto entry_vars = [&](int e, int& a, int& bb) {
          bb = -1;
          if (e < 64) a = 3 + (e >> 3), bb = 3 + (e & 7);
          else if (e < 88) a = 3 + (e - 64) / 3, bb = (e - 64) % 3;
          else if (e < 97) a = (e - 88) / 3, bb = (e - 88) % 3;
          else a = e - 97;
        };
        auto live = [&](int a) { return a < 3 || (a - 3 < n && ((ancg >> (a - 3)) & 1u)); };
        {  // joint-joint block and joint gradient: moments of the points x
          double W21[21], v6[6];
          goal_gram_moments(rb, ga, Y, W21, v6);
          for (int e = l; e < 108; e += 32) {
            int a, bb;
            entry_vars(e, a, bb);
            if (!(e < 64 || e >= 100)) continue;
            double v = 0.0;
            if (live(a) && (bb < 0 || live(bb))) {
              double sa[6], sb[6];
              base_screw(a, scr, bxp, byp, sa);
              if (bb < 0) {
#pragma unroll
                for (int q = 0; q < 6; ++q) v += sa[q] * v6[q];
              } else {
                base_screw(bb, scr, bxp, byp, sb);
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                  double u = 0.0;
#pragma unroll
                  for (int q = 0; q < 6; ++q) u += W21[sym6(p, q)] * sb[q];
                  v += sa[p] * u;
                }
              }
            }
            sys[e] = v;
          }
        }
        {  // base-base block and base gradient: moments of the target points tau; v6 = sum X_tau^T (tau - x)
          double W21[21], v6[6];
          goal_gram_moments(rb, Y, ga, W21, v6);
          for (int e = l; e < 108; e += 32) {
            int a, bb;
            entry_vars(e, a, bb);
            if (!(e >= 88 && e < 100)) continue;
            double v = 0.0, sa[6], sb[6];
            base_screw(a, scr, bxp, byp, sa);
            if (bb < 0) {
#pragma unroll
              for (int q = 0; q < 6; ++q) v += sa[q] * v6[q];
            } else {
              base_screw(bb, scr, bxp, byp, sb);
#pragma unroll
              for (int p = 0; p < 6; ++p) {
                double u = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) u += W21[sym6(p, q)] * sb[q];
                v += sa[p] * u;
              }
            }
            sys[e] = v;
          }
        }
        if (l < 24) {  // coupling block: C[j][a] = -s_j^T (sum X_x^T X_tau) sigma_a
          double Wc[36];
          goal_cross_moments(rb, ga, Y, Wc);
          const int e = 64 + l;
          int a, bb;
          entry_vars(e, a, bb);
          double v = 0.0;
          if (live(a)) {
            double sa[6], sb[6];
            base_screw(a, scr, bxp, byp, sa);
            base_screw(bb, scr, bxp, byp, sb);
#pragma unroll
            for (int p = 0; p < 6; ++p) {
              double u = 0.0;
#pragma unroll
              for (int q = 0; q < 6; ++q) u += Wc[6 * p + q] * sb[q];
              v -= sa[p] * u;
            }
          }
          sys[e] = v;
        }
        if (l == 0) sys[108] = fi;
      }
      wave_sync();
    }
    __syncthreads();
    if (tid == 0) {
      double ft = w_effort * (bxp * bxp + byp * byp + th * th);
      for (int i = 0; i < ng; ++i) ft += s_sys[((size_t)ts_ * n_max + i) * GTO_BASE_SYS + 108];
      s_red[0] = ft;
    }
    __syncthreads();
    const double f_try = s_red[0];
    // ---- accept / reject (block-uniform)
    int done = 0, take = 0;
    if (first) {
      first = 0;
      take = 1;
      f = f_try;
    } else if (f_try < f && pred > 0.0) {
      const double df = f - f_try, rho = df / pred;
      f = f_try;
      slot = 1 - slot;
      take = 1;
      const double sg = 2.0 * rho - 1.0;
      double fac = 1.0 - sg * sg * sg;
      fac = fmax(fac, 1.0 / 3.0);
      lambda = fmax(lambda * fac, 1e-12);
      nu = 2.0;
      if (df <= sp.tol_rel_f * (1.0 + f)) {
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
    if (take)
      for (int i = tid; i < NV; i += 256) s_x[i] = s_xt[i];
    if (done) break;
    if (k >= sp.max_iter) {
      status = GTO_STATUS_MAX_ITER;
      break;
    }
    __syncthreads();
    // ---- step at the current iterate.  (1) base gradient half and base active set
    const double* sysc = s_sys + (size_t)slot * n_max * GTO_BASE_SYS;
    if (tid < 3) {
      double g = w_effort * s_x[tid];
      for (int i = 0; i < ng; ++i) g += sysc[(size_t)i * GTO_BASE_SYS + 97 + tid];
      const double lo = tid == 2 ? -PI_ : -INFINITY, hi = tid == 2 ? PI_ : INFINITY;
      s_red[4 + tid] = g;
      s_red[8 + tid] = ((s_x[tid] <= lo && g > 0.0) || (s_x[tid] >= hi && g < 0.0)) ? 1.0 : 0.0;
    }
    if (tid == 3) s_red[1] = 0.0;  // failure flag
    __syncthreads();
    const bool acty0 = s_red[8] != 0.0, acty1 = s_red[9] != 0.0, acty2 = s_red[10] != 0.0;
    // (2) eliminate the goal blocks: E_i = D_i^-1 C_i, u_i = D_i^-1 rhs_i, and their Schur contributions
    for (int i = wave; i < ng; i += 4) {
      const double* sys = sysc + (size_t)i * GTO_BASE_SYS;
      const double* xi = s_x + 8 + 8 * i;
      const double xr = xi[r], xc = xi[c], br = sys[100 + r], bc = sys[100 + c];
      const int rr = r < n ? r : 0, cc = c < n ? c : 0;
      const bool ar = r >= n || (xr <= rb->lower[rr] && br > 0.0) || (xr >= rb->upper[rr] && br < 0.0);
      const bool ac = c >= n || (xc <= rb->lower[cc] && bc > 0.0) || (xc >= rb->upper[cc] && bc < 0.0);
      double S = sys[lane];
      if (ar || ac) S = (r == c) ? 1.0 : 0.0;
      else if (r == c) S *= (1.0 + lambda);
      const int bad = gj_invert8(S, lane, r, c);
      const double C0 = (ac || acty0) ? 0.0 : sys[64 + 3 * c], C1 = (ac || acty1) ? 0.0 : sys[64 + 3 * c + 1],
                   C2 = (ac || acty2) ? 0.0 : sys[64 + 3 * c + 2];
      const double e0 = matvec8(S, C0), e1 = matvec8(S, C1), e2 = matvec8(S, C2), u = matvec8(S, ac ? 0.0 : -bc);
      if (c == 0) {
        s_E[i * 24 + 3 * r] = e0, s_E[i * 24 + 3 * r + 1] = e1, s_E[i * 24 + 3 * r + 2] = e2;
        s_u[i * 8 + r] = u;
      }
      if (__any(bad) && lane == 0) s_red[1] = 1.0;
      wave_sync();
      if (lane < 12) {
        const int a = lane < 9 ? lane / 3 : lane - 9, a2 = lane < 9 ? lane % 3 : -1;
        const bool acta = a == 0 ? acty0 : (a == 1 ? acty1 : acty2);
        double v = 0.0;
        for (int j = 0; j < n; ++j) {
          const double xj = xi[j], bj = sys[100 + j];
          const bool aj = (xj <= rb->lower[j] && bj > 0.0) || (xj >= rb->upper[j] && bj < 0.0);
          const double cj = (aj || acta) ? 0.0 : sys[64 + 3 * j + a];
          v += cj * (a2 >= 0 ? s_E[i * 24 + 3 * j + a2] : s_u[i * 8 + j]);
        }
        s_part[i * 16 + lane] = v;
      }
    }
    __syncthreads();
    // (3) 3x3 Schur complement in goal order, Cholesky, base step
    if (tid == 0) {
      double M[9], rh[3];
      const bool act[3] = {acty0, acty1, acty2};
      for (int a = 0; a < 3; ++a) {
        for (int a2 = 0; a2 < 3; ++a2) {
          double v = (a == a2) ? w_effort : 0.0;
          for (int i = 0; i < ng; ++i) v += sysc[(size_t)i * GTO_BASE_SYS + 88 + 3 * a + a2];
          if (act[a] || act[a2]) v = (a == a2) ? 1.0 : 0.0;
          else if (a == a2) v *= (1.0 + lambda);
          M[3 * a + a2] = v;
        }
        rh[a] = act[a] ? 0.0 : -s_red[4 + a];
      }
      for (int i = 0; i < ng; ++i) {
        for (int e = 0; e < 9; ++e) M[e] -= s_part[i * 16 + e];
        for (int a = 0; a < 3; ++a) rh[a] -= s_part[i * 16 + 9 + a];
      }
      // Cholesky M = L L^T
      int bad = 0;
      double L00 = M[0], L10, L20, L11, L21, L22;
      if (!(L00 > 0.0)) bad = 1;
      L00 = sqrt(L00);
      L10 = M[3] / L00, L20 = M[6] / L00;
      L11 = M[4] - L10 * L10;
      if (!(L11 > 0.0)) bad = 1;
      L11 = sqrt(L11);
      L21 = (M[7] - L20 * L10) / L11;
      L22 = M[8] - L20 * L20 - L21 * L21;
      if (!(L22 > 0.0)) bad = 1;
      L22 = sqrt(L22);
      const double z0 = rh[0] / L00, z1 = (rh[1] - L10 * z0) / L11, z2 = (rh[2] - L20 * z0 - L21 * z1) / L22;
      const double d2 = z2 / L22, d1 = (z1 - L21 * d2) / L11, d0 = (z0 - L10 * d1 - L20 * d2) / L00;
      s_red[12] = d0, s_red[13] = d1, s_red[14] = d2;
      if (bad) s_red[1] = 1.0;
    }
    __syncthreads();
    if (s_red[1] != 0.0) {
      status = GTO_STATUS_NUMERICAL;
      break;
    }
    // (4) back-substitute, project, and the terms of the predicted reduction
    const double dy0 = s_red[12], dy1 = s_red[13], dy2 = s_red[14];
    const double sy0 = dy0, sy1 = dy1, sy2 = fmin(fmax(s_x[2] + dy2, -PI_), PI_) - s_x[2];
    if (tid == 0) {
      s_xt[0] = s_x[0] + dy0, s_xt[1] = s_x[1] + dy1, s_xt[2] = fmin(fmax(s_x[2] + dy2, -PI_), PI_);
    }
    for (int i = wave; i < ng; i += 4) {
      const double* sys = sysc + (size_t)i * GTO_BASE_SYS;
      const double* xi = s_x + 8 + 8 * i;
      const int rr = r < n ? r : 0;
      const double dq = s_u[i * 8 + r] - (s_E[i * 24 + 3 * r] * dy0 + s_E[i * 24 + 3 * r + 1] * dy1 + s_E[i * 24 + 3 * r + 2] * dy2);
      double v = fmin(fmax(xi[r] + dq, rb->lower[rr]), rb->upper[rr]);
      if (r >= n) v = 0.0;
      const double sr = v - xi[r];
      const double sc_ = __shfl(sr, c << 3, 64);
      if (c == 0) s_xt[8 + 8 * i + r] = v;
      double part = sys[lane] * sr * sc_;  // s_i^T D_i s_i
      if (c < 3) {
        const double syc = c == 0 ? sy0 : (c == 1 ? sy1 : sy2);
        part += 2.0 * sys[64 + 3 * r + c] * sr * syc;  // 2 s_i^T C_i s_y
      }
      if (c == 3) part += 2.0 * sys[100 + r] * sr;  // 2 g_q . s_i
      if (lane < 9) {
        const int a = lane / 3, a2 = lane % 3;
        const double sa = a == 0 ? sy0 : (a == 1 ? sy1 : sy2), sb = a2 == 0 ? sy0 : (a2 == 1 ? sy1 : sy2);
        part += sys[88 + lane] * sa * sb;  // s_y^T S_i s_y
      }
      if (lane >= 9 && lane < 12) part += 2.0 * sys[97 + lane - 9] * (lane == 9 ? sy0 : (lane == 10 ? sy1 : sy2));
      part = wave_sum(part);
      double ms = (c == 0 && r < n) ? fabs(sr) : 0.0;
      ms = wave_max(ms);
      if (lane == 0) {
        s_part[i * 16 + 12] = part;
        s_part[i * 16 + 13] = ms;
      }
    }
    __syncthreads();
    if (tid == 0) {
      double q = w_effort * (sy0 * sy0 + sy1 * sy1 + sy2 * sy2) + 2.0 * w_effort * (s_x[0] * sy0 + s_x[1] * sy1 + s_x[2] * sy2);
      double ms = fmax(fabs(sy0), fmax(fabs(sy1), fabs(sy2)));
      for (int i = 0; i < ng; ++i) {
        q += s_part[i * 16 + 12];
        ms = fmax(ms, s_part[i * 16 + 13]);
      }
      s_red[2] = -q;
      s_red[3] = ms;
    }
    __syncthreads();
    if (s_red[3] < sp.tol_step) {
      status = GTO_STATUS_CONVERGED;
      break;
    }
    pred = s_red[2];
  }
  __syncthreads();
  if (tid < 3) y_out[(size_t)b * 3 + tid] = s_x[tid];
  for (int idx = tid; idx < n_max * ndof; idx += 256) {
    const int i = idx / ndof, dq = idx % ndof, j = rb->opt_of_dof[dq];
    q_out[((size_t)b * n_max + i) * ndof + dq] = (j >= 0 && i < ng) ? s_x[8 + 8 * i + j] : s_qc[dq];
  }
  if (tid == 0) {
    if (cost_out) cost_out[b] = f;
    if (iters_out) iters_out[b] = k;
    if (status_out) status_out[b] = status;
  }
}

// ------------------------------------------------------------------------------------------------
// Step kernel for robots with 9..16 optimised joints (mobile manipulators: BASELINE configs[4]): the same accept / reject,
// active set, block-tridiagonal solve, projected step and predicted decrease as k_lm_step, written for width, not for
// latency.  One workgroup of 256 threads per slot; thread (r, c) = (tid >> 4, tid & 15) owns entry (r, c) of the 16x16
// blocks (a 16-lane DPP row is a matrix row: mat-vec products reduce with row_sum16).  The block recursion runs forward
// over all waypoints (Gauss-Jordan without pivoting on the SPD blocks, pivot row / column exchanged through LDS), the
// inverses Z_s go to a workspace in HBM (L2) and come back for the back substitution.
//   dynamic LDS (doubles): Q [NP][T] | b, y, x [m][NP] each | e [m][NP] | S [2][NP*NP] | red [32] ; int act [m]
__host__ __device__ inline size_t lm_wide_lds_bytes(int T, int NP) {
  const size_t m = (size_t)T - 2;
  return ((size_t)NP * T + 4 * m * NP + 2 * NP * NP + 32) * sizeof(double) + (m + 8) * sizeof(int);
}

template <int NP>
__global__ __launch_bounds__(256) void k_lm_step_wide(const RobotDev* __restrict__ rb, BatchPtrs bp, SolveParams sp, int B,
                                                      double* __restrict__ Zws /* [slots][T-2][NP*NP] */) {
  static_assert(NP == 16, "one thread per entry of a 16x16 block");
  typedef Blk<NP> BK;
  const int tid = threadIdx.x, lane = tid & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0 && bp.progress) {  // lagged by design: what had finished when this launch started
    const int nd = __hip_atomic_load(bp.n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(bp.progress, bp.progress_tag | (unsigned long long)(unsigned)nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const bool listed = bp.slot_inst != nullptr;
  const int slot_id = blockIdx.x;
  const int b = listed ? bp.slot_inst[slot_id] : slot_id;
  if (b < 0) return;  // empty slot
  __shared__ int s_nid;
  InstState* st = bp.state + b;
  if (st->done) return;
  extern __shared__ __attribute__((aligned(16))) double smem_w[];
  const int T = sp.T, n = rb->n_opt, m = T - 2, nF = rb->n_frames;
  double* s_Q = smem_w;                      // [NP][T]
  double* s_b = s_Q + NP * T;                // [m][NP]
  double* s_y = s_b + m * NP;
  double* s_x = s_y + m * NP;
  double* s_e = s_x + m * NP;
  double* s_S = s_e + m * NP;                // [2][NP*NP] pivot exchange
  double* s_red = s_S + 2 * NP * NP;         // [32]
  int* s_act = reinterpret_cast<int*>(s_red + 32);  // [m] frozen-joint bit masks
  const int r = tid >> 4, c = tid & 15;
  const int trial = 1 - st->slot;
  double* __restrict__ qfb = listed ? bp.qfs + (size_t)slot_id * T * nF : bp.qf + (size_t)b * T * nF;
  double* __restrict__ Zb = Zws + (size_t)slot_id * m * NP * NP;

  // ---- P0: objective of the trial point
  double fo = 0.0;
  {
    const double* blk = bp.blocks + ((size_t)trial * B + b) * T * BK::STRIDE;
    for (int t = 2 + lane; t < T; t += 64) fo += blk[(size_t)t * BK::STRIDE + BK::SS];
    fo = wave_sum(fo);
    fo += bp.ss_fixed[4 * b] + bp.ss_fixed[4 * b + 1];
  }
  const double f_try = st->fgoal_try + sp.w_obstacle * fo + st->fvel_try;
  // ---- P1: accept / reject (uniform over the workgroup)
  double f = st->f, lambda = st->lambda, nu = st->nu;
  int slot = st->slot, done = 0, status = st->status;
  const int k = st->evals;
  const int argmin_try = st->argmin_try, argmin_cur0 = st->argmin_cur;
  const double pred0 = st->pred;
  bool accept = false;
  if (st->first) {
    accept = true;
  } else if (f_try < f && pred0 > 0.0) {
    accept = true;
    const double df = f - f_try, rho = df / pred0;
    const double sg = 2.0 * rho - 1.0;
    double fac = 1.0 - sg * sg * sg;
    fac = fmax(fac, 1.0 / 3.0);
    lambda = fmax(lambda * fac, 1e-12);
    nu = 2.0;
    if (df <= sp.tol_rel_f * (1.0 + f_try)) {
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
  double* __restrict__ Qc = bp.Qcur + (size_t)b * n * T;
  double* __restrict__ Qt = bp.Qtry + (size_t)b * n * T;
  if (accept) {
    f = f_try;
    slot = trial;
  }
  const int argmin_cur = accept ? argmin_try : argmin_cur0;
  for (int idx = tid; idx < NP * T; idx += 256) {
    const double v = idx < n * T ? (accept ? Qt : Qc)[idx] : 0.0;
    s_Q[idx] = v;
    if (accept && idx < n * T) Qc[idx] = v;
  }
  if (!done && k >= sp.max_iter) {
    status = GTO_STATUS_MAX_ITER;
    done = 1;
  }
#define GTO_FINISH_W(STATUS)                      \
  do {                                            \
    if (tid == 0) {                               \
      st->f = f;                                  \
      st->lambda = lambda;                        \
      st->nu = nu;                                \
      st->slot = slot;                            \
      st->first = 0;                              \
      st->done = 1;                               \
      st->status = (STATUS);                      \
      st->argmin_cur = argmin_cur;                \
      atomicAdd(bp.n_done, 1);                    \
      if (listed) {                               \
        const int nid = atomicAdd(bp.next, 1);    \
        s_nid = nid < bp.n_total ? nid : -1;      \
        bp.slot_inst[slot_id] = s_nid;            \
      }                                           \
    }                                             \
    if (listed) {                                 \
      __syncthreads();                            \
      const int nid = s_nid;                      \
      if (nid >= 0)                               \
        for (int i_ = tid; i_ < sp.T * nF; i_ += 256)     \
          bp.qfs[(size_t)slot_id * sp.T * nF + i_] = bp.qf[(size_t)nid * sp.T * nF + i_]; \
    }                                             \
    return;                                       \
  } while (0)
  __syncthreads();
  if (done) GTO_FINISH_W(status);

  // ---- P2: gradient b = J^T r at the current iterate, active set, right-hand side
  const double* __restrict__ oblk = bp.blocks + ((size_t)slot * B + b) * T * BK::STRIDE;
  const double* __restrict__ gblk = bp.goalblk + ((size_t)slot * B + b) * 2 * BK::STRIDE;
  const double alpha = sp.alpha;
  for (int s = tid; s < m; s += 256) s_act[s] = 0;
  __syncthreads();
  for (int idx = tid; idx < m * NP; idx += 256) {
    const int sI = idx / NP, i = idx % NP, t = sI + 2;
    double bv = 0.0;
    int act = 1;  // padded rows count as frozen
    if (i < n) {
      bv = sp.w_obstacle * oblk[(size_t)t * BK::STRIDE + BK::JTR + i];
      if (t == T - 1) bv += gblk[BK::JTR + i];
      if (sp.use_standoff && t == sp.ts) bv += gblk[BK::STRIDE + BK::JTR + i];
      const double qt = s_Q[i * T + t], qm = s_Q[i * T + t - 1];
      bv += alpha * (qt - qm);
      if (t < T - 1) bv -= alpha * (s_Q[i * T + t + 1] - qt);
      act = (qt <= rb->lower[i] && bv > 0.0) || (qt >= rb->upper[i] && bv < 0.0);
    }
    s_b[idx] = bv;
    if (act) atomicOr(&s_act[sI], 1 << i);
  }
  __syncthreads();
  for (int idx = tid; idx < m * NP; idx += 256) {
    const int sI = idx / NP, i = idx % NP;
    const int a0 = (s_act[sI] >> i) & 1;
    const int a1 = (sI < m - 1) ? (s_act[sI + 1] >> i) & 1 : 1;
    s_e[idx] = (a0 || a1) ? 0.0 : -alpha;
    s_y[idx] = a0 ? 0.0 : -s_b[idx];
  }
  __syncthreads();
  // undamped block entry (r, c) of waypoint s: obstacle + goal + velocity terms
  const bool inb = r < n && c < n;
  auto a_entry = [&](int s) {
    const int t = s + 2;
    double a = inb ? sp.w_obstacle * oblk[(size_t)t * BK::STRIDE + BK::JTJ + tid] : 0.0;
    if (inb && r == c) a += (t == T - 1) ? alpha : 2.0 * alpha;
    if (inb && t == T - 1) a += gblk[BK::JTJ + tid];
    if (inb && sp.use_standoff && t == sp.ts) a += gblk[BK::STRIDE + BK::JTJ + tid];
    return a;
  };
  // ---- P3: forward block recursion S_s = D_s - E Z_{s-1} E, Z_s = S_s^-1, y_s = Z_s (rhs_s - E y_{s-1})
  int bad = 0;
  {
    double Zprev = 0.0;
    double a_next = a_entry(0);
    for (int s = 0; s < m; ++s) {
      const double a = a_next;
      if (s + 1 < m) a_next = a_entry(s + 1);  // the next block's loads fly during this block's inversion
      const int am = s_act[s];
      const bool frozen = ((am >> r) & 1) || ((am >> c) & 1);
      double S = frozen ? (r == c ? 1.0 : 0.0) : (r == c ? a * (1.0 + lambda) : a);
      double zc = s_y[s * NP + c];
      if (s > 0) {
        const double er = s_e[(s - 1) * NP + r], ec = s_e[(s - 1) * NP + c];
        S -= er * ec * Zprev;
        zc -= ec * s_x[(s - 1) * NP + c];  // s_x holds y during the forward sweep
      }
      // Gauss-Jordan, no pivoting (SPD): the pivot row and column go through LDS, two buffers, one barrier per pivot
#pragma unroll 1
      for (int j = 0; j < NP; ++j) {
        double* buf = s_S + (j & 1) * NP * NP;
        buf[tid] = S;
        __syncthreads();
        const double pjj = buf[j * NP + j], prj = buf[r * NP + j], pjc = buf[j * NP + c];
        if (!(pjj > 0.0)) bad = 1;
        const double piv = 1.0 / pjj;
        const double tcol = -prj * piv;
        const double in_row = (c == j) ? piv : pjc * piv;
        const double off_row = (c == j) ? tcol : fma(tcol, pjc, S);
        S = (r == j) ? in_row : off_row;
      }
      Zb[(size_t)s * NP * NP + tid] = S;
      Zprev = S;
      const double yr = row_sum16(S * zc);
      __syncthreads();  // every thread has read y_{s-1} before it is needed no more... and the pivot buffers are free
      if (c == 0) s_x[s * NP + r] = yr;
      __syncthreads();
    }
  }
  if (tid == 0) s_red[0] = 0.0;
  __syncthreads();
  if (bad) s_red[0] = 1.0;
  __syncthreads();
  if (s_red[0] != 0.0) GTO_FINISH_W(GTO_STATUS_NUMERICAL);
  // ---- back substitution x_s = y_s - Z_s E_s x_{s+1}
  {
    double zn = m >= 2 ? Zb[(size_t)(m - 2) * NP * NP + tid] : 0.0;
    for (int s = m - 2; s >= 0; --s) {
      const double Z = zn;
      if (s > 0) zn = Zb[(size_t)(s - 1) * NP * NP + tid];
      const double pr = row_sum16(Z * (s_e[s * NP + c] * s_x[(s + 1) * NP + c]));
      __syncthreads();
      if (c == 0) s_x[s * NP + r] -= pr;
      __syncthreads();
    }
  }
  // ---- P4: projected trial point; s_x becomes the projected step
  double maxstep = 0.0;
  for (int idx = tid; idx < m * NP; idx += 256) {
    const int sI = idx / NP, i = idx % NP, t = sI + 2;
    double sv = 0.0;
    if (i < n) {
      const double q0 = s_Q[i * T + t];
      double v = q0 + s_x[idx];
      v = fmin(fmax(v, rb->lower[i]), rb->upper[i]);
      Qt[(size_t)i * T + t] = v;
      qfb[(size_t)t * nF + rb->opt_frame[i]] = v;  // the obstacle kernel reads joint values by frame
      sv = v - q0;
    }
    s_x[idx] = sv;
    maxstep = fmax(maxstep, fabs(sv));
  }
  if (tid < n * 2) Qt[(size_t)(tid >> 1) * T + (tid & 1)] = s_Q[(tid >> 1) * T + (tid & 1)];
  maxstep = wave_max(maxstep);
  if (lane == 0) s_red[1 + (tid >> 6)] = maxstep;
  __syncthreads();
  maxstep = fmax(fmax(s_red[1], s_red[2]), fmax(s_red[3], s_red[4]));
  if (maxstep < sp.tol_step) GTO_FINISH_W(GTO_STATUS_CONVERGED);
  // ---- P5: predicted decrease of the undamped model: -(2 b.s + s^T A s)
  {
    double part = 0.0;
    for (int s = 0; s < m; ++s) {
      const double sr = s_x[s * NP + r], scv = s_x[s * NP + c];
      double v = a_entry(s) * sr * scv;
      if (c == 0) {
        const double xn = (s < m - 1) ? s_x[(s + 1) * NP + r] : 0.0;
        v += 2.0 * sr * fma(-alpha, xn, s_b[s * NP + r]);
      }
      part += v;
    }
    part = wave_sum(part);
    if (lane == 0) s_red[8 + (tid >> 6)] = part;
  }
  __syncthreads();
  const double acc = (s_red[8] + s_red[9]) + (s_red[10] + s_red[11]);
  if (tid == 0) {
    st->f = f;
    st->lambda = lambda;
    st->nu = nu;
    st->pred = -acc;
    st->slot = slot;
    st->first = 0;
    st->status = status;
    st->evals = k + 1;
    st->argmin_cur = argmin_cur;
  }
#undef GTO_FINISH_W
}

__global__ __launch_bounds__(64) void k_lm_finalize(const RobotDev* __restrict__ rb, BatchPtrs bp, SolveParams sp, int B,
                                                    double* Q_out, double* dQ_out, double* cost_out,
                                                    int32_t* iters_out, int32_t* status_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int T = sp.T, n = rb->n_opt, ndof = rb->ndof;
  const InstState* st = bp.state + b;
  const double* Q0b = bp.Q0 + (size_t)b * ndof * T;
  const double* Qc = bp.Qcur + (size_t)b * n * T;
  if (Q_out) {
    double* Qo = Q_out + (size_t)b * ndof * T;
    for (int idx = lane; idx < ndof * T; idx += 64) Qo[idx] = Q0b[idx];  // parameter rows (optas/solver.py:151-153)
    __syncthreads();
    for (int idx = lane; idx < n * T; idx += 64) Qo[(size_t)rb->opt_index[idx / T] * T + idx % T] = Qc[idx];
  }
  if (dQ_out) {
    double* dQo = dQ_out + (size_t)b * ndof * (T - 1);
    for (int idx = lane; idx < ndof * (T - 1); idx += 64) dQo[idx] = 0.0;
    __syncthreads();
    for (int idx = lane; idx < n * (T - 2); idx += 64) {
      const int j = idx / (T - 2), t = 1 + idx % (T - 2);
      dQo[(size_t)rb->opt_index[j] * (T - 1) + t] = (Qc[(size_t)j * T + t + 1] - Qc[(size_t)j * T + t]) / sp.dt;
    }
  }
  if (lane == 0) {
    if (cost_out) cost_out[b] = st->f;
    if (iters_out) iters_out[b] = st->evals;
    if (status_out) status_out[b] = st->status;
  }
}

// ------------------------------------------------------------------------------------------------
// Evaluation kernels (parity / seed scoring entry points)
#define GTO_EVAL_TG 4  // configurations per workgroup of k_eval_kin
__host__ __device__ inline int eval_kin_lds_doubles(int F, int L, int n) {
  return fk_tab_doubles(F, L, n) + GTO_EVAL_TG * F * 2 + fk_scratch_doubles(F, GTO_EVAL_TG) + GTO_EVAL_TG * L * 12 + GTO_EVAL_TG * GTO_NB * 6;
}
// Kinematics of nq configurations by the solver's own forward kinematics (fk_mfma_tree, four configurations per
// workgroup): frames_out [nq][F][16] global transform of every frame (4x4 row-major; gto_eval_fk, pinned against
// optas/models.py:826-868) and / or vis_out [nq][L][12] visual transforms of the collision links (gto_eval_points).
__global__ __launch_bounds__(256) void k_eval_kin(const RobotDev* __restrict__ rb, int nq, const double* __restrict__ q,
                                                  double* __restrict__ frames_out, double* __restrict__ vis_out) {
  constexpr int TG = GTO_EVAL_TG;
  const int i0 = blockIdx.x * TG, tid = threadIdx.x;
  const int F = rb->n_frames, L = rb->n_links, n = rb->n_opt, ndof = rb->ndof;
  const int ng = min(TG, nq - i0);
  extern __shared__ __attribute__((aligned(16))) double smem_ek[];
  double* s_tab = smem_ek;
  double* s_sc = s_tab + fk_tab_doubles(F, L, n);
  double* s_X = s_sc + TG * F * 2;
  double* s_vis = s_X + fk_scratch_doubles(F, TG);
  double* s_screw = s_vis + TG * L * 12;
  const int nt = fk_tab_doubles(F, L, n);
  for (int k = tid; k < nt; k += 256) s_tab[k] = rb->fk_tab[k];
  for (int idx = tid; idx < ng * F; idx += 256) {
    const int kq = idx / F, f = idx - kq * F;
    const int jt = rb->joint_type[f], dq = rb->q_index[f];
    double a = 0.0, cs = 1.0;
    if (dq >= 0) {
      const double qv = q[(size_t)(i0 + kq) * ndof + dq];
      if (jt == GTO_JOINT_REVOLUTE) sincos(qv, &a, &cs);
      else if (jt == GTO_JOINT_PRISMATIC) a = qv;
    }
    s_sc[2 * idx] = a;
    s_sc[2 * idx + 1] = cs;
  }
  __syncthreads();
  fk_mfma_tree(rb, s_tab, ng, s_sc, s_X, reinterpret_cast<int*>(s_X + ng * 32 * F + 64), tid, s_vis, s_screw);
  __syncthreads();
  if (frames_out) {  // X_f = G_f^T, row-major, in the ping-pong half the last round wrote
    for (int idx = tid; idx < ng * F * 16; idx += 256) {
      const int kq = idx / (F * 16), r_ = idx - kq * F * 16, f = r_ >> 4, e = r_ & 15;
      const double* Xg = s_X + (size_t)kq * 32 * F + (rb->fk_rounds & 1) * 16 * F;
      double v = Xg[16 * f + 4 * (e & 3) + (e >> 2)];
      if (e >= 12) v = (e == 15) ? 1.0 : 0.0;  // the affine part is exact by construction
      frames_out[((size_t)(i0 + kq) * F + f) * 16 + e] = v;
    }
  }
  if (vis_out)
    for (int idx = tid; idx < ng * L * 12; idx += 256) vis_out[(size_t)i0 * L * 12 + idx] = s_vis[idx];
}

// thread per (configuration, sorted point); outputs in the caller's original point order via perm
__global__ void k_eval_points(const RobotDev* __restrict__ rb, const double* __restrict__ px, const double* __restrict__ py,
                              const double* __restrict__ pz, const int32_t* __restrict__ plink,
                              const int32_t* __restrict__ perm, const SceneDev* __restrict__ scene, int nq,
                              const double* __restrict__ vis, const double* __restrict__ base, int use_obs,
                              double* xyz_out, int32_t* off_out, double* val_out, double* grad_out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int iq = blockIdx.y;
  const int P = rb->n_points;
  if (p >= P) return;
  const double* V = vis + ((size_t)iq * rb->n_links + plink[p]) * 12;
  const double x0 = px[p], x1 = py[p], x2 = pz[p];
  const double X0 = V[0] * x0 + V[1] * x1 + V[2] * x2 + V[3] + base[3 * iq];
  const double X1 = V[4] * x0 + V[5] * x1 + V[6] * x2 + V[7] + base[3 * iq + 1];
  const double X2 = V[8] * x0 + V[9] * x1 + V[10] * x2 + V[11] + base[3 * iq + 2];
  const size_t o = (size_t)iq * P + perm[p];
  if (xyz_out) {
    xyz_out[3 * o] = X0;
    xyz_out[3 * o + 1] = X1;
    xyz_out[3 * o + 2] = X2;
  }
  if (!scene) return;
  const SceneDev sc = *scene;
  const float* field = use_obs ? sc.c_obs : sc.c_all;
  const int ix = voxel_axis(X0, sc.ox, sc.res, sc.rinv, sc.nx);
  const int iy = voxel_axis(X1, sc.oy, sc.res, sc.rinv, sc.ny);
  const int iz = voxel_axis(X2, sc.oz, sc.res, sc.rinv, sc.nz);
  const int off = iz + sc.nz * (iy + sc.ny * ix);
  if (off_out) off_out[o] = off;
  if (val_out) val_out[o] = (double)field[off];
  if (grad_out) {
    const int ixp = min(ix + 1, sc.nx - 1), ixm = max(ix - 1, 0);
    const int iyp = min(iy + 1, sc.ny - 1), iym = max(iy - 1, 0);
    const int izp = min(iz + 1, sc.nz - 1), izm = max(iz - 1, 0);
    grad_out[3 * o] = ((double)field[iz + sc.nz * (iy + sc.ny * ixp)] - (double)field[iz + sc.nz * (iy + sc.ny * ixm)]) * sc.inv2r;
    grad_out[3 * o + 1] = ((double)field[iz + sc.nz * (iyp + sc.ny * ix)] - (double)field[iz + sc.nz * (iym + sc.ny * ix)]) * sc.inv2r;
    grad_out[3 * o + 2] = ((double)field[izp + sc.nz * (iy + sc.ny * ix)] - (double)field[izm + sc.nz * (iy + sc.ny * ix)]) * sc.inv2r;
  }
}

// compute_plan_cost (gto/gto_models.py:204-215): block per (plan, waypoint), plain sum of c_obs
#define GTO_PLAN_TG 4  // waypoints per workgroup of k_plan_cost
__host__ __device__ inline int plan_cost_lds_doubles(int F, int L, int n) {
  return fk_tab_doubles(F, L, n) + GTO_PLAN_TG * F * 2 + fk_scratch_doubles(F, GTO_PLAN_TG) + GTO_PLAN_TG * L * 12 + GTO_PLAN_TG * GTO_NB * 6 + 4 * GTO_PLAN_TG;
}
// One workgroup per (plan, group of four waypoints): kinematics of the four configurations on the matrix cores
// (fk_mfma_tree, as in the obstacle kernel), then every thread walks its surface points once for all four waypoints.
// The per-waypoint sums are formed in the order thread-strided partial sums -> wave -> waves 0..3.
__global__ __launch_bounds__(256) void k_plan_cost(const RobotDev* __restrict__ rb, const double* __restrict__ px,
                                                   const double* __restrict__ py, const double* __restrict__ pz,
                                                   const int32_t* __restrict__ plink, const SceneDev* __restrict__ scene,
                                                   int T, const double* __restrict__ plans, const double* __restrict__ base,
                                                   double* __restrict__ partial /*[n][T]*/) {
  constexpr int TG = GTO_PLAN_TG;
  const int t0 = blockIdx.x * TG, i = blockIdx.y, tid = threadIdx.x;
  const int F = rb->n_frames, L = rb->n_links, n = rb->n_opt, ndof = rb->ndof;
  const int ng = min(TG, T - t0);
  extern __shared__ __attribute__((aligned(16))) double smem_pc[];
  double* s_tab = smem_pc;
  double* s_sc = s_tab + fk_tab_doubles(F, L, n);  // [TG][F][2]
  double* s_X = s_sc + TG * F * 2;
  double* s_vis = s_X + fk_scratch_doubles(F, TG);  // [TG][L][12]
  double* s_screw = s_vis + TG * L * 12;            // [TG][GTO_NB][6] (by-product of the kinematics, unused here)
  double* s_red = s_screw + TG * GTO_NB * 6;        // [4][TG]
  const int nt = fk_tab_doubles(F, L, n);
  for (int k = tid; k < nt; k += 256) s_tab[k] = rb->fk_tab[k];
  for (int idx = tid; idx < ng * F; idx += 256) {
    const int kq = idx / F, f = idx - kq * F;
    const int jt = rb->joint_type[f], dq = rb->q_index[f];
    double a = 0.0, cs = 1.0;
    if (dq >= 0) {
      const double qv = plans[((size_t)i * ndof + dq) * T + t0 + kq];
      if (jt == GTO_JOINT_REVOLUTE) sincos(qv, &a, &cs);
      else if (jt == GTO_JOINT_PRISMATIC) a = qv;
    }
    s_sc[2 * idx] = a;
    s_sc[2 * idx + 1] = cs;
  }
  __syncthreads();
  fk_mfma_tree(rb, s_tab, ng, s_sc, s_X, reinterpret_cast<int*>(s_X + ng * 32 * F + 64), tid, s_vis, s_screw);
  __syncthreads();
  const SceneDev sc = *scene;
  const double b0 = base[0], b1 = base[1], b2 = base[2];
  double acc[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g) acc[g] = 0.0;
  for (int p = tid; p < rb->n_points; p += 256) {
    const double x0 = px[p], x1 = py[p], x2 = pz[p];
    const int l = plink[p];
#pragma unroll
    for (int g = 0; g < TG; ++g) {
      if (g < ng) {
        const double* V = s_vis + (g * L + l) * 12;
        const int ix = voxel_axis(V[0] * x0 + V[1] * x1 + V[2] * x2 + V[3] + b0, sc.ox, sc.res, sc.rinv, sc.nx);
        const int iy = voxel_axis(V[4] * x0 + V[5] * x1 + V[6] * x2 + V[7] + b1, sc.oy, sc.res, sc.rinv, sc.ny);
        const int iz = voxel_axis(V[8] * x0 + V[9] * x1 + V[10] * x2 + V[11] + b2, sc.oz, sc.res, sc.rinv, sc.nz);
        acc[g] += (double)sc.c_obs[iz + sc.nz * (iy + sc.ny * ix)];
      }
    }
  }
#pragma unroll
  for (int g = 0; g < TG; ++g) {
    const double v = wave_sum(acc[g]);
    if ((tid & 63) == 0) s_red[(tid >> 6) * TG + g] = v;
  }
  __syncthreads();
  if (tid < ng) partial[(size_t)i * T + t0 + tid] = ((s_red[tid] + s_red[TG + tid]) + s_red[2 * TG + tid]) + s_red[3 * TG + tid];
}

// ------------------------------------------------------------------------------------------------
// Cost field from a depth image (mesh_to_sdf/depth_point_cloud.py:9-141): the producer
// of the (F,) cost arrays.  Arithmetic follows the reference's order with FMA contraction switched off
// in these two kernels: the reference values are reproduced bit for bit (tests/golden/depth_cost.npz);
// where the reference's BLAS products could differ in the last bit on another machine, the CPU
// restatement (oracle, -ffp-contract=off) is matched exactly.
// backproject (:32-52) + world transform (:21-23); invalid pixels become points at infinity
__global__ void k_depth_backproject(const float* __restrict__ depth, int H, int W, const double* __restrict__ Kinv,
                                    const double* __restrict__ cam, const uint8_t* __restrict__ target_mask,
                                    double threshold, double* __restrict__ px, double* __restrict__ py,
                                    double* __restrict__ pz, uint8_t* __restrict__ valid) {
#pragma clang fp contract(off)  // plain operators below must stay unfused (HIP's __dmul_rn & co. are no barrier)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const float d = depth[i];
  const bool ok = (d > 0.0f) && ((double)d < threshold) && (!target_mask || target_mask[i] == 0);
  double X[3], P[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double t = (Kinv[3 * r] * (double)x + Kinv[3 * r + 1] * (double)y) + Kinv[3 * r + 2];
    X[r] = (double)d * t;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    P[r] = ((cam[4 * r] * X[0] + cam[4 * r + 1] * X[1]) + cam[4 * r + 2] * X[2]) + cam[4 * r + 3];
  px[i] = ok ? P[0] : INFINITY;
  py[i] = ok ? P[1] : INFINITY;
  pz[i] = ok ? P[2] : INFINITY;
  valid[i] = ok ? 1 : 0;
}

// get_sdf (:56-61) + is_outside (:126-141) + cost map (:84-89): one query per thread, the cloud streamed
// through LDS in tiles; exact nearest neighbour by exhaustive search in FP64 (the KD-tree of the reference
// returns the same distance), a few ms for 10^5 queries x 3*10^5 points at the FP64 vector rate.
// What follows the nearest-neighbour search of a query (mesh_to_sdf/depth_point_cloud.py:56-141): sign by the depth-buffer
// visibility test, cost map.  best = squared distance to the nearest point of the cloud.
__device__ __forceinline__ void depth_sdf_finish(bool live, long q, double q0, double q1, double q2, double best,
                                                 const float* __restrict__ depth, int H, int W, const double* __restrict__ K,
                                                 const double* __restrict__ cam_inv, float epsilon, float w_inside,
                                                 float* __restrict__ sdf_out, uint8_t* __restrict__ inside_out,
                                                 float* __restrict__ cost_out) {
#pragma clang fp contract(off)
  if (!live) return;
  float dist = (float)sqrt(best);
  double pc[3], u[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    pc[r] = ((cam_inv[4 * r] * q0 + cam_inv[4 * r + 1] * q1) + cam_inv[4 * r + 2] * q2) + cam_inv[4 * r + 3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    u[r] = (K[3 * r] * pc[0] + K[3 * r + 1] * pc[1]) + K[3 * r + 2] * pc[2];
  const double ux = u[0] / u[2], uy = u[1] / u[2];
  // .astype(int): truncation toward zero; non-finite / out-of-range values become INT64_MIN in NumPy
  const bool fx = fabs(ux) < 9.0e18, fy = fabs(uy) < 9.0e18;
  const long ix = fx ? (long)ux : LONG_MIN, iy = fy ? (long)uy : LONG_MIN;
  bool outside = true;
  if (ix >= 0 && iy >= 0 && ix < W && iy < H) outside = pc[2] < (double)depth[iy * W + ix];
  if (!outside) dist = -dist;
  float c = 0.0f;
  if (!outside) {
    c = w_inside * (-dist + epsilon / 2.0f);
  } else if (dist > 0.0f && dist < epsilon) {
    const float e = dist - epsilon;
    c = (e * e) / (2.0f * epsilon);
  }
  if (sdf_out) sdf_out[q] = dist;
  if (inside_out) inside_out[q] = outside ? 0 : 1;
  if (cost_out) cost_out[q] = c;
}

__global__ __launch_bounds__(256) void k_depth_sdf(const double* __restrict__ px, const double* __restrict__ py,
                                                   const double* __restrict__ pz, int N, const float* __restrict__ depth,
                                                   int H, int W, const double* __restrict__ K,
                                                   const double* __restrict__ cam_inv, const double* __restrict__ query,
                                                   long nq, float epsilon, float w_inside, float* __restrict__ sdf_out,
                                                   uint8_t* __restrict__ inside_out, float* __restrict__ cost_out) {
#pragma clang fp contract(off)  // see k_depth_backproject
  __shared__ double sx[256], sy[256], sz[256];
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = q < nq;
  const double q0 = live ? query[3 * q] : 0.0, q1 = live ? query[3 * q + 1] : 0.0, q2 = live ? query[3 * q + 2] : 0.0;
  double best = INFINITY;
  for (int base = 0; base < N; base += 256) {
    const int j = base + threadIdx.x;
    sx[threadIdx.x] = j < N ? px[j] : INFINITY;
    sy[threadIdx.x] = j < N ? py[j] : INFINITY;
    sz[threadIdx.x] = j < N ? pz[j] : INFINITY;
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 256; ++k) {
      const double dx = q0 - sx[k], dy = q1 - sy[k], dz = q2 - sz[k];
      const double d2 = (dx * dx + dy * dy) + dz * dz;
      best = fmin(best, d2);  // NaN (inf - inf never occurs: queries are finite) is ignored by fmin
    }
    __syncthreads();
  }
  depth_sdf_finish(live, q, q0, q1, q2, best, depth, H, W, K, cam_inv, epsilon, w_inside, sdf_out, inside_out, cost_out);
}

// ---- the same nearest-neighbour distances without the exhaustive search.  The cloud comes from a depth image, so
// pixels that are close in the image are (mostly) close in space: tiles of 8 x 4 pixels are the leaves of a bounding-box
// hierarchy, the tiles taken in Morton order of their (column, row) so that every node of the implicit complete binary
// tree (heap indexing, P x P leaf slots, P a power of two) covers a rectangle of the image.  No sorting, no copy of
// the points.  A query walks the tree nearer child first and skips every box that cannot hold a closer point.  The
// skip test is exact in floating point: for a point p of a box, |q - p| >= (distance of q to the box) holds per axis
// also after rounding (subtraction, product and sum are monotone), and the box distance is summed in the same order
// as the point distance, so the minimum over the visited points is the minimum over all points, bit for bit.
#define GTO_BVH_TILE_W 8
#define GTO_BVH_TILE_H 4
__host__ __device__ inline unsigned bvh_compact1by1(unsigned v) {
  v &= 0x55555555u;
  v = (v | (v >> 1)) & 0x33333333u;
  v = (v | (v >> 2)) & 0x0f0f0f0fu;
  v = (v | (v >> 4)) & 0x00ff00ffu;
  v = (v | (v >> 8)) & 0x0000ffffu;
  return v;
}
// boxes: [node][6] = lo x, y, z, hi x, y, z; an empty box is (+inf, -inf): its distance from anything is +inf
__global__ void k_bvh_leaves(const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ pz, int H,
                             int W, int P, double* __restrict__ boxes) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= P * P) return;
  const int cx = (int)bvh_compact1by1((unsigned)s), cy = (int)bvh_compact1by1((unsigned)s >> 1);
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int r = 0; r < GTO_BVH_TILE_H; ++r)
    for (int c = 0; c < GTO_BVH_TILE_W; ++c) {
      const int y = cy * GTO_BVH_TILE_H + r, x = cx * GTO_BVH_TILE_W + c;
      if (y >= H || x >= W) continue;
      const size_t j = (size_t)y * W + x;
      const double v[3] = {px[j], py[j], pz[j]};
      if (!(v[0] < INFINITY)) continue;  // invalid pixel (point at infinity)
      for (int k = 0; k < 3; ++k) {
        lo[k] = fmin(lo[k], v[k]);
        hi[k] = fmax(hi[k], v[k]);
      }
    }
  double* b = boxes + (size_t)(P * P - 1 + s) * 6;
  for (int k = 0; k < 3; ++k) b[k] = lo[k], b[3 + k] = hi[k];
}
// inner nodes, level by level from the leaves up (one workgroup: 2 P^2 nodes are a few ten thousand)
__global__ __launch_bounds__(1024) void k_bvh_up(int P, double* __restrict__ boxes) {
  for (int first = (P * P - 1) / 2, count = P * P / 2; count >= 1; first = (first - 1) / 2, count >>= 1) {
    for (int i = threadIdx.x; i < count; i += 1024) {
      const int n = first + i;
      const double* a = boxes + (size_t)(2 * n + 1) * 6;
      const double* c = boxes + (size_t)(2 * n + 2) * 6;
      double* o = boxes + (size_t)n * 6;
      for (int k = 0; k < 3; ++k) o[k] = fmin(a[k], c[k]), o[3 + k] = fmax(a[3 + k], c[3 + k]);
    }
    __threadfence_block();
    __syncthreads();
    if (count == 1) break;
  }
}
__device__ __forceinline__ double bvh_box_d2(const double* __restrict__ b, double q0, double q1, double q2) {
#pragma clang fp contract(off)
  const double ex = fmax(fmax(b[0] - q0, q0 - b[3]), 0.0), ey = fmax(fmax(b[1] - q1, q1 - b[4]), 0.0),
               ez = fmax(fmax(b[2] - q2, q2 - b[5]), 0.0);
  return (ex * ex + ey * ey) + ez * ez;
}
// Queries are visited in Morton order of their position (30-bit keys over the cloud's bounding box grown by its own
// extent on every side; the sort is hipCUB's radix sort), so that the 64 lanes of a wave ask for neighbouring points
// and walk nearly the same boxes: in the caller's order (a grid in C order: 64 consecutive voxels are a line across
// the whole workspace) the lanes of a wave diverge at every node.
__host__ __device__ inline unsigned bvh_part1by2(unsigned v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ void k_query_keys(const double* __restrict__ query, long nq, const double* __restrict__ boxes, unsigned* __restrict__ keys,
                             unsigned* __restrict__ idx) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  unsigned key = 0u;
  for (int k = 0; k < 3; ++k) {
    const double lo = boxes[k], hi = boxes[3 + k], ext = hi - lo;
    double u = ext > 0.0 && ext < INFINITY ? (query[3 * q + k] - (lo - ext)) / (3.0 * ext) : 0.0;
    u = fmin(fmax(u, 0.0), 1.0);
    key |= bvh_part1by2((unsigned)(u * 1023.0)) << k;
  }
  keys[q] = key;
  idx[q] = (unsigned)q;
}
__global__ __launch_bounds__(256) void k_depth_sdf_bvh(const double* __restrict__ px, const double* __restrict__ py,
                                                       const double* __restrict__ pz, const double* __restrict__ boxes, int P,
                                                       const unsigned* __restrict__ order,
                                                       const float* __restrict__ depth, int H, int W,
                                                       const double* __restrict__ K, const double* __restrict__ cam_inv,
                                                       const double* __restrict__ query, long nq, float epsilon, float w_inside,
                                                       float* __restrict__ sdf_out, uint8_t* __restrict__ inside_out,
                                                       float* __restrict__ cost_out, unsigned long long* __restrict__ stats) {
#pragma clang fp contract(off)
  // PACKET traversal: the 64 queries of a wave (neighbours in space, see k_query_keys) walk the tree TOGETHER with one
  // stack; a node is entered when any lane still needs it, and the 32 points of a leaf are fetched once per wave and
  // tried by every lane.  Trying more points than a lane needs cannot change its minimum (they are points of the cloud),
  // so the result is the exhaustive search's; what changes is that a leaf costs one coalesced read per wave instead of
  // one scattered read per lane (per-lane traversal moved 18 KB per query through the caches).
  __shared__ int s_stack[4][64];
  __shared__ double s_sbox[4][64][6];  // the box of every stacked node (read from memory once, when its parent is entered)
  __shared__ double s_pts[4][3][GTO_BVH_TILE_W * GTO_BVH_TILE_H];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long slot = (long)blockIdx.x * 256 + tid;
  const bool live = slot < nq;
  const long q = live ? (long)order[slot] : 0;  // the slot-th query in Morton order
  const double q0 = live ? query[3 * q] : 0.0, q1 = live ? query[3 * q + 1] : 0.0, q2 = live ? query[3 * q + 2] : 0.0;
  double best = INFINITY;
  const int first_leaf = P * P - 1;
  int* stk = s_stack[wave];
  int sp = 0;  // wave-uniform
  unsigned n_pop = 0, n_leaf = 0;
  if (lane == 0) stk[0] = 0;
  if (lane < 6) s_sbox[wave][0][lane] = boxes[lane];
  sp = 1;
  wave_sync_lds();
  while (sp > 0) {
    const int n = __builtin_amdgcn_readfirstlane(stk[sp - 1]);
    --sp;
    ++n_pop;
    const bool need = live && bvh_box_d2(s_sbox[wave][sp], q0, q1, q2) < best;
    if (!__any(need)) continue;  // too far for every lane (it may have become so since it was pushed)
    if (n >= first_leaf) {
      ++n_leaf;
      const int s = n - first_leaf;
      const int cx = (int)bvh_compact1by1((unsigned)s), cy = (int)bvh_compact1by1((unsigned)s >> 1);
      if (lane < GTO_BVH_TILE_W * GTO_BVH_TILE_H) {  // one pixel per lane; pixels outside the image count as invalid
        const int y = cy * GTO_BVH_TILE_H + lane / GTO_BVH_TILE_W, x = cx * GTO_BVH_TILE_W + lane % GTO_BVH_TILE_W;
        const bool in = y < H && x < W;
        const size_t j = in ? (size_t)y * W + x : 0;
        s_pts[wave][0][lane] = in ? px[j] : INFINITY;
        s_pts[wave][1][lane] = in ? py[j] : INFINITY;
        s_pts[wave][2][lane] = in ? pz[j] : INFINITY;
      }
      wave_sync_lds();
#pragma unroll 8
      for (int k = 0; k < GTO_BVH_TILE_W * GTO_BVH_TILE_H; ++k) {
        const double dx = q0 - s_pts[wave][0][k], dy = q1 - s_pts[wave][1][k], dz = q2 - s_pts[wave][2][k];
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        best = fmin(best, d2);  // invalid pixels are points at infinity: d2 = inf
      }
      wave_sync_lds();  // every lane is done with the tile before the next leaf overwrites it
    } else {
      const int c1 = 2 * n + 1, c2 = c1 + 1;
      const double bx = lane < 12 ? boxes[(size_t)c1 * 6 + lane] : 0.0;  // both children's boxes: twelve consecutive doubles
      const double d1 = bvh_box_d2(boxes + (size_t)c1 * 6, q0, q1, q2), d2 = bvh_box_d2(boxes + (size_t)c2 * 6, q0, q1, q2);
      const bool n1 = live && d1 < best, n2 = live && d2 < best;
      // the child more lanes are closer to is entered first (it is pushed last)
      const bool c1_first = __popcll(__ballot(live && d1 <= d2)) * 2 >= __popcll(__ballot(live));
      const bool any1 = __any(n1), any2 = __any(n2);
      const int firstc = c1_first ? c1 : c2, secondc = c1_first ? c2 : c1;
      const bool any_first = c1_first ? any1 : any2, any_second = c1_first ? any2 : any1;
      // lane l < 12 holds entry l % 6 of child c1 (l < 6) or c2: it files it under the slot its child gets
      const bool mine_is_second = (lane < 6) != c1_first;
      if (any_second) {
        if (lane == 0) stk[sp] = secondc;
        if (lane < 12 && mine_is_second) s_sbox[wave][sp][lane % 6] = bx;
        ++sp;
      }
      if (any_first) {
        if (lane == 0) stk[sp] = firstc;
        if (lane < 12 && !mine_is_second) s_sbox[wave][sp][lane % 6] = bx;
        ++sp;
      }
      wave_sync_lds();
    }
  }
  if (stats) {
    if (lane == 0) {
      atomicAdd(stats, (unsigned long long)n_pop * 64);
      atomicAdd(stats + 1, (unsigned long long)n_leaf * 64);
      atomicAdd(stats + 2, (unsigned long long)n_pop);
    }
  }
  depth_sdf_finish(live, q, q0, q1, q2, best, depth, H, W, K, cam_inv, epsilon, w_inside, sdf_out, inside_out, cost_out);
}
