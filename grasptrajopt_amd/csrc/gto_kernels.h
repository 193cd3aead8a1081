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
#include <type_traits>

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

// Candidate trial points per instance and round (speculation): after a rejected evaluation the next
// trial point is known without looking at anything new (same iterate, same normal equations, damping lambda * nu), so
// the step kernel may hand out the first few members of that chain at once and the obstacle kernel evaluates them in one
// launch; the next step kernel walks them in order and stops at the first one the sequential rule accepts.  Same
// iterates, same iteration counts as one candidate per round (oracle/gto_oracle.c solve_instance), fewer dependent rounds.
#define GTO_KSPEC 4
// Issue priority of the step kernels' wavefronts (s_setprio, 0..3): a step workgroup is one dependent chain per instance
// and sits on its lane's critical path, while the obstacle launches of the OTHER lanes fill the same SIMDs with five
// waves of throughput work each; measured with four lanes in flight (rocprofv3 kernel trace, profiles/r06_*): the
// step launch takes 3-4x its duration alone (configs[4]: 450 us against 123; configs[2]: 120 against 38).
#ifndef GTO_STEP_PRIO
#define GTO_STEP_PRIO 0
#endif
// job-list entry: instance id in the low 24 bits, the block set the evaluation writes to above it, the candidate above
// that (-1: void).  The set rides in the entry so that the obstacle kernel needs nothing of the instance's state.
#define GTO_JOB_SHIFT 28
#define GTO_JOB_SET_SHIFT 24
#define GTO_JOB_MASK 0x00ffffff
#define GTO_JOB_ENTRY(b, set, cand) ((b) | ((set) << GTO_JOB_SET_SHIFT) | ((cand) << GTO_JOB_SHIFT))
#define GTO_JOB_SET(le) (((le) >> GTO_JOB_SET_SHIFT) & 15)

// entries behind a parity's item list: an itemized obstacle launch is laid out over 8 ceil(jobs / 8) groups-per-job
// workgroups, up to seven jobs' worth of groups more than the list can hold, and each reads its entry before it compares
// its index with the list's length
#define GTO_ITEM_SLACK (8 * GTO_MAX_T)
struct InstState {
  double f, lambda, nu;              // lambda, nu: the damping candidate 0 was generated with
  double pred[GTO_KSPEC];            // predicted decrease of candidate j
  double fgoal_try[GTO_KSPEC], fvel_try[GTO_KSPEC];
  int32_t argmin_try[GTO_KSPEC];
  int32_t slot, first, done, status, evals, argmin_cur;
  int32_t ncand;                     // candidates generated by the last step (1 .. GTO_KSPEC)
  int32_t cflags;                    // bit j: the solve of candidate j >= 1 failed numerically; bit 8 + j: its step is below tol_step;
                                     // bits 16..23: rounds in a row in which the FIRST candidate was accepted (saturating)
  unsigned long long nz0, nz1;       // non-zero blocks of the current iterate's set: bit s = waypoint s + 2 (nz0), s + 66 (nz1)
  // emptiness certificates (launches with few instances in flight; BatchPtrs::room): room_ok: the block set of the current
  // iterate has a room for every (waypoint, link); cand_rooms: so will the sets of the candidates generated last, once they
  // have been evaluated (the step that generated them certified or listed every group, and the launch that evaluates the
  // listed ones leaves their rooms)
  int32_t room_ok, cand_rooms;
};

struct SolveParams {
  int32_t T, ts, use_standoff, n_max, max_iter, grad_mode;
  int32_t interleave;  // obstacle kernel: waypoints of a group nG apart (1) instead of consecutive (0); same results
  int32_t dbg_cut;  // debug: leave the obstacle kernel after phase k (1 prologue, 2 broad phase, 3 loop); 0 = off
  // solve loop (rounds): `round` counts the rounds of the call, `parity` = round & 1 selects the live list the kernels of
  // this round read (the step kernel fills the other one); kcap = candidate copies the workspace holds per instance (the
  // block sets number kcap + 1); a step launch generates k_acc candidates after an accepted evaluation and k_rej after a
  // round without one; k_eval = candidates per position the obstacle launch is laid out for
  int32_t round, parity, kcap, k_acc, k_rej, k_eval;
  // one-candidate launches (k_lm_step<4,1>): an instance KEEPS its position in the lists from round to round (the instance
  // that takes over from a finished one inherits it; a position nobody takes is void) instead of drawing a new one from
  // the lists' counters: hundreds of workgroups adding to one address are served one by one
  int32_t static_pos;
  // an instance whose first candidate was accepted spec_streak rounds in a row (a run of good steps: the next one is
  // unlikely to be rejected) gets one candidate after the next accepted evaluation instead of k_acc; 0 = never
  int32_t spec_streak;
  // Broad phase ahead of the obstacle launch (the rounds that fill the GPU; prebroad_tail): the step kernel tests the
  // bounding spheres of its new trial trajectory itself, settles the waypoint groups none of whose spheres can reach a
  // non-zero voxel record and lists the others; the next obstacle launch is laid out over that list.  pb_tg / pb_ng:
  // waypoints per group and groups per job of that layout; pb_pw: waypoints per pass of the tail (what its LDS holds);
  // pb_mC: division magic of the chunk count; pb_verify (debug builds): every group is listed, the settled ones marked
  int32_t pb_next, pb_tg, pb_ng, pb_pw, pb_verify;
  // Emptiness certificates ahead of the obstacle launch (launches with FEW instances in flight, k_lm_step<8, 4>): the step
  // kernel settles the waypoint groups of its candidates whose links all keep room (cert_tail) and lists the others, over
  // groups of their own (one waypoint each: the heavy waypoints near the goal then sit in workgroups of their own)
  int32_t cert_next, cert_tg, cert_ng;  // (cert_tg consecutive waypoints per group of the certificates' item list, cert_ng groups per job)
  int32_t pb_C, pb_tab0, pb_npar;  // chunks of moving links; first double of the tail's tables in LDS (PbLayout); actuated joints that are not optimised
  uint32_t pb_mC, pb_mF;  // division magics of the chunk count and of the frame count
  double pb_eps;          // metres by which the culling radius is widened (transforms stored in single precision)
  double dt, alpha, w_obstacle, w_vel, tol_step, tol_rel_f, lambda0;
};

#define GTO_NLIVE(p) (2 * (p))
#define GTO_NJOBS(p) (2 * (p) + 1)
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
  double* Qtry;      // [kcap][B][n][T] candidate trial trajectories
  double* blocks;    // [kcap + 1][B][T][BLK_STRIDE]: set `slot` holds the current iterate's, candidate j goes to set (slot + 1 + j) mod (kcap + 1)
  double* goalblk;   // [kcap + 1][B][2][BLK_STRIDE]
  double* ss_fixed;  // [B][4]  sum c^2 of the two pinned waypoints; of the static links under c_all / c_obs
  int32_t* n_done;   // [1]     instances that have finished
  double* qf;        // [B][T][F] joint value of every frame of the trial trajectory (0 for fixed joints)
  // solve loop: at most `cap` instances are in flight.  live[p][i], i < nlive[p], are the instances the step kernel of a
  // round of parity p works on (one workgroup per position i); jobs[p][q], q < njobs[p], are the evaluations the obstacle
  // kernel of that round does: one per (instance, candidate trial point), entry = instance | candidate << GTO_JOB_SHIFT
  // (-1: void).  The step kernel of an instance that goes on appends it and its candidates to the lists of the other
  // parity, the step kernel of one that finishes appends the next instance of the call that has not started (*next = its
  // id, n_total ids in all), so the lists stay dense and the host sizes the launches by the instances that can still be
  // in flight.  The joint values of a job's trajectory live in qfs, indexed by parity and job: the obstacle kernel needs
  // no instance id to start its kinematics.  live == null outside the solve loop: the kernels then index the batch directly.
  int32_t* live;       // [2][cap]
  int32_t* jobs;       // [2][cap * kcap]
  // [4], eight-byte aligned: nlive[2 p] instances, nlive[2 p + 1] jobs of the lists of parity p -- a pair, so that a workgroup
  // of the step kernel takes its places in both lists with ONE 64-bit add (GTO_NLIVE, GTO_NJOBS)
  int32_t* nlive;
  int32_t* next;       // [1]
  double* qfs;         // [2][cap * kcap][T][F]
  // Sparse outputs of an evaluation inside the solve loop: one 64-byte record per (set, b, t), written
  // whole by whoever settles the waypoint (a partially written line costs the memory side a read-modify-write): [0] the
  // sum of c^2 of waypoint t, [1] != 0 iff the block (set, b, t) holds a non-zero J^T J / J^T r -- only those blocks are
  // written, and only those are read back by the step kernel (nine tenths of the blocks of a table-top workload are zeros).
  double* wrec;        // [kcap + 1][B][T][8]
  // items[p][i], i < nitems[p] (= nlive[8 + p]): the (job, group) pairs the obstacle kernel of the round of parity p has
  // to look at (the others were settled by the step kernel's broad phase): .x = the job's list entry, .y = job index << 8 | group.
  int2* items;         // [2][cap * kcap * groups]
  // Emptiness certificates: room[set][b][t][l] = metres every point of link l at waypoint t of the trajectory whose
  // evaluation block set `set` holds may still move before a bounding sphere of the link could reach a non-zero voxel
  // record (< 0: none / a sphere survives; 1e30: a link that is never tested).  Written by the obstacle launch that looks
  // at the (waypoint, link) -- (d - R - 1) voxels of the tightest culled sphere, in metres -- and by the step kernel for
  // the groups it certifies (what is left after the candidate's step).  Null: no certificates in this call.
  float* room;         // [kcap + 1][B][T][L]
  const SceneDev* scenes;
  int32_t cap, n_total;
  // lanes of one call (gto_api.hip): this lane's lists serve the instances b0 .. n_total - 1 of the batch, the first w0 of
  // them in flight from round 0 on (one lane: b0 = 0, w0 = cap)
  int32_t b0, w0;
  long long* dbg;    // optional: phase timestamps of instance 0's step kernel (GTO_DEBUG_TIMING)
  // progress word in pinned host memory (or null): the first workgroup of every step launch publishes
  // (call tag << 32 | instances finished so far); the host polls it instead of copying n_done back through the stream
  // (call tag << 32 | instances finished so far) in word 0 and (call tag << 32 | round of this launch) in word 1
  unsigned long long* progress;
  unsigned long long progress_tag;
  unsigned long long* work;  // [64] or null: surface points gathered (one voxel record or field value each), in 64 cells by blockIdx
};

// ------------------------------------------------------------------------------------------------
// Pointers that arrive through a descriptor in memory (SceneDev) are GENERIC to the compiler: it loads through them with
// flat_load, which counts on the LDS counter as well as on the vector-memory counter, so every wait for an LDS result
// also waits for the gathers in flight (and the other way round) -- the software pipelines of the gather loops were
// serialised by their own list appends.  Every scene array lives in device memory: say so.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* as_global(const T* p) {
  return (const __attribute__((address_space(1))) T*)p;
}
// x clipped to [0, hi] in one instruction (the compiler does not know 0 <= hi and keeps a max and a min)
__device__ __forceinline__ int clamp_index(int x, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}
__device__ __forceinline__ double4 load_record(const VoxelRec* base, unsigned off) {
  typedef double gto_rec4 __attribute__((ext_vector_type(4)));
  const gto_rec4 v = *(const __attribute__((address_space(1))) gto_rec4*)(base + off);
  return make_double4(v.x, v.y, v.z, v.w);
}

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
__host__ __device__ inline int fk_tab_doubles(int F, int L, int n) { return GTO_FK_STRIDE * F + 16 * L + 16 * n + L + 2 * n + F; }
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
// NB configurations go through every stage side by side: their LDS reads are issued together, then their matrix
// instructions, then their stores (a wave alone on its SIMD otherwise pays the LDS round trip and the matrix-core latency
// once per configuration: 1.1-1.7 K cycles a round for three of them).  Members past the end of a batch repeat its last
// configuration and store to the dummy target: no branch inside a stage, same operations and bits for the real ones.
template <int NB>
__device__ __forceinline__ void fk_mfma_tree_nb(const RobotDev* __restrict__ rb, const double* __restrict__ s_tab, int ng,
                                       const double* __restrict__ s_sc, double* __restrict__ s_X, int* __restrict__ s_anc,
                                       int tid, double* __restrict__ s_vis, double* __restrict__ s_screw,
                                       long long* dbgp, int scr_np, int F, int rounds) {
  // `ng` configurations are advanced together, stage by stage, so that they share the barriers:
  // s_sc [ng][F][2], s_X [ng][2][F][16] then a dummy store target [64], s_anc [2][GTO_MAX_FRAMES] (the
  // ancestors do not depend on the configuration), s_vis [ng][L][12], s_screw [ng][scr_np][6]
  // (F frames, `rounds` rounds of pointer jumping: the full tree of RobotDev::fk_tab or the compact one of fk_tab_c)
  const int L = rb->n_links, n = rb->n_opt;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the compiler must see it as wave-uniform
  const int ra = lane >> 4, rc = lane & 3, blk = (lane >> 2) & 3;
  const int e = 4 * ra + rc, et = 4 * rc + ra;
  const double ident = (ra == rc) ? 1.0 : 0.0;
  const int nGf = (F + 3) >> 2;
  double* s_dummy = s_X + ng * 32 * F;  // target of the stores of blocks that have nothing to say
  const double* tVo = s_tab + GTO_FK_STRIDE * F;
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
    const double* kt = s_tab + GTO_FK_STRIDE * f;
    const int es = e ^ (5 * (f & 3));  // bank placement of the frame's entries (gto_device.h, fkx)
    const double c0 = kt[16 + es], c1 = kt[32 + es], K = kt[48 + es], Ot = kt[es];  // the table holds the origin transposed
    areg[k] = (int)tI[L + 2 * n + f];
    for (int kq0 = 0; kq0 < ng; kq0 += NB) {
      double sn[NB], cs[NB], X[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int kq = min(kq0 + j, ng - 1);
        sn[j] = s_sc[2 * (kq * F + f)], cs[j] = s_sc[2 * (kq * F + f) + 1];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) X[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(fma(sn[j], K, fma(cs[j], c1, c0)), Ot, 0.0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NB; ++j) *((f0 < F && kq0 + j < ng) ? s_X + (kq0 + j) * 32 * F + fkx(f, e) : s_dummy + lane) = X[j];
    }
    s_anc[f] = areg[k];  // sixteen lanes, one value
  }
  __syncthreads();
  if (dbgp && tid == 0) dbgp[1] = clock64();
  int cur = 0;
  for (int rd = 0; rd < rounds; ++rd) {
    const int* Ac = s_anc + cur * GTO_MAX_FRAMES;
    int* Aw = s_anc + (1 - cur) * GTO_MAX_FRAMES;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const int g = wave + 4 * k;
      if (g >= nGf) continue;
      const int f0 = 4 * g + blk, f = f0 < F ? f0 : F - 1;
      const int a = areg[k], ac = a >= 0 ? a : 0;
      const int a2x = Ac[ac];
      for (int kq0 = 0; kq0 < ng; kq0 += NB) {
        double Aop[NB], Bx[NB], X[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const double* Xc = s_X + min(kq0 + j, ng - 1) * 32 * F + cur * 16 * F;
          Aop[j] = Xc[fkx(f, et)];  // A[i][k] = X_f[i][k]
          Bx[j] = Xc[fkx(ac, e)];   // B[k][j] = X_anc[k][j]
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) X[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(Aop[j], a >= 0 ? Bx[j] : ident, 0.0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) *((f0 < F && kq0 + j < ng) ? s_X + (kq0 + j) * 32 * F + (1 - cur) * 16 * F + fkx(f, e) : s_dummy + lane) = X[j];
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
      const double Vo = tVo[fkx(l, e)];
      const int fl = (int)tI[l];
      for (int kq0 = 0; kq0 < ng; kq0 += NB) {
        double Xg[NB], V[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) Xg[j] = (s_X + min(kq0 + j, ng - 1) * 32 * F + cur * 16 * F)[fkx(fl, e)];  // X_f = G_f^T, row-major
#pragma unroll
        for (int j = 0; j < NB; ++j) V[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(Vo, Xg[j], 0.0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) *((l1 < L && rc < 3 && kq0 + j < ng) ? s_vis + ((kq0 + j) * L + l) * 12 + 4 * rc + ra : s_dummy + lane) = V[j];
      }
    } else {
      // lanes 0-15 of a block row hold a = R u, lanes 16-31 o = frame origin
      const int j1 = 4 * (og - nGl) + blk, jo = j1 < n ? j1 : n - 1;
      const bool prism = tI[L + n + jo] != 0.0;
      const double U = tU[fkx(jo, e)];
      const int fj = (int)tI[L + jo];
      for (int kq0 = 0; kq0 < ng; kq0 += NB) {
        double Xg[NB], S[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) Xg[j] = (s_X + min(kq0 + j, ng - 1) * 32 * F + cur * 16 * F)[fkx(fj, e)];
#pragma unroll
        for (int j = 0; j < NB; ++j) S[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(U, Xg[j], 0.0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const bool w = j1 < n && ra == 0 && rc < 3 && kq0 + j < ng;
          const double av = S[j], ov = __shfl(S[j], (lane + 16) & 63, 64);
          const double a1 = quad_perm<1, 2, 0, 3>(av), a2 = quad_perm<2, 0, 1, 3>(av);
          const double o1 = quad_perm<1, 2, 0, 3>(ov), o2 = quad_perm<2, 0, 1, 3>(ov);
          const double cr = o1 * a2 - o2 * a1;
          double* sv = s_screw + ((kq0 + j) * scr_np + jo) * 6;
          *(w ? sv + rc : s_dummy + lane) = prism ? 0.0 : av;
          *(w ? sv + 3 + rc : s_dummy + lane) = prism ? av : cr;
        }
      }
    }
  }
  if (dbgp && tid == 0) dbgp[7] = clock64();
}

__device__ __forceinline__ void fk_mfma_tree(const RobotDev* __restrict__ rb, const double* __restrict__ s_tab, int ng,
                                    const double* __restrict__ s_sc, double* __restrict__ s_X, int* __restrict__ s_anc,
                                    int tid, double* __restrict__ s_vis, double* __restrict__ s_screw,
                                    long long* dbgp = nullptr, int scr_np = GTO_NB) {
  fk_mfma_tree_nb<1>(rb, s_tab, ng, s_sc, s_X, s_anc, tid, s_vis, s_screw, dbgp, scr_np, rb->n_frames, rb->fk_rounds);
}
// ... of the compact tree (RobotDev::fk_tab_c; s_sc and s_X are laid out for its n_cframes frames)
__device__ __forceinline__ void fk_mfma_tree_compact(const RobotDev* __restrict__ rb, const double* __restrict__ s_tab, int ng,
                                            const double* __restrict__ s_sc, double* __restrict__ s_X, int* __restrict__ s_anc,
                                            int tid, double* __restrict__ s_vis, double* __restrict__ s_screw,
                                            long long* dbgp, int scr_np, int Fc, int rounds_c) {
  fk_mfma_tree_nb<1>(rb, s_tab, ng, s_sc, s_X, s_anc, tid, s_vis, s_screw, dbgp, scr_np, Fc, rounds_c);
}
// The batched form, for launches whose workgroups have their SIMDs to themselves (the obstacle kernel's variant for few
// instances in flight): the kinematics of a group of two take 5.0 K cycles instead of 6.5 K.  With the GPU full the
// one-at-a-time form is 1.7 % faster (same-box A/B), so it stays everywhere else.
__device__ __forceinline__ void fk_mfma_tree_batched(const RobotDev* __restrict__ rb, const double* __restrict__ s_tab, int ng,
                                            const double* __restrict__ s_sc, double* __restrict__ s_X, int* __restrict__ s_anc,
                                            int tid, double* __restrict__ s_vis, double* __restrict__ s_screw,
                                            long long* dbgp, int scr_np, int F, int rounds) {
  if (ng == 1) fk_mfma_tree_nb<1>(rb, s_tab, ng, s_sc, s_X, s_anc, tid, s_vis, s_screw, dbgp, scr_np, F, rounds);
  else if (ng == 2) fk_mfma_tree_nb<2>(rb, s_tab, ng, s_sc, s_X, s_anc, tid, s_vis, s_screw, dbgp, scr_np, F, rounds);
  else fk_mfma_tree_nb<3>(rb, s_tab, ng, s_sc, s_X, s_anc, tid, s_vis, s_screw, dbgp, scr_np, F, rounds);
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

// Launch geometry of k_obstacle_gram, worked out by the host: the kernel's entry then holds no integer division (25
// scalar instructions each, five of them) and no load from the robot table; 31 % of the kernel's scalar instructions and
// 23 % of its wave-cycles were spent before the first useful load (PMC passes with the kernel cut after its entry).
struct ObsGeom {
  int32_t F, L, n, C, TG, cap_active, nt;  // frames of the COMPACT tree, links, optimised joints, chunks; waypoints per group; doubles of its FK table
  int32_t Fq, rounds;                      // frames of the robot (stride of the joint-value tables); rounds of pointer jumping of the compact tree
  int32_t nG, q_nT, r_nT;                  // waypoint groups per job; nT = q_nT * nG + r_nT
  uint32_t m_nG, m_F, m_C, m_L;            // ceil(2^32 / d): x / d == __umulhi(x, m) while x d < 2^32 (d == 1: x itself)
  ObsLds lay;
  ObsGeom(int F_, int Fq_, int rounds_, int L_, int n_, int C_, int TG_, int nT, int NP)
      : F(F_), L(L_), n(n_), C(C_), TG(TG_), cap_active(TG_ * C_), nt(fk_tab_doubles(F_, L_, n_)), Fq(Fq_), rounds(rounds_), lay(TG_, F_, L_, TG_ * C_, NP) {
    nG = (nT + TG - 1) / TG;
    q_nT = nT / nG, r_nT = nT % nG;
    m_nG = magic(nG), m_F = magic(F), m_C = magic(C), m_L = magic(L);
  }
  static uint32_t magic(int d) { return (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); }
};
__device__ __forceinline__ int fast_div(int x, int d, uint32_t m) { return d == 1 ? x : (int)__umulhi((uint32_t)x, m); }

struct InstState;
template <int NP>
__device__ __forceinline__ void trial_goal_terms_wave(const RobotDev* rb, const BatchPtrs& bp, const SolveParams& sp, int B,
                                             int b, int lane, int trial, int cand, InstState* st, double* s_q, double* s_fr,
                                             double* s_gaff, double* s_gscr);

#ifndef GTO_OBS_MIN_WAVES
#define GTO_OBS_MIN_WAVES 5  // waves per SIMD the register allocator must leave room for: five workgroups per CU (31 KB of LDS each at three waypoints per workgroup)
#endif
// PD: chunks of a wave whose record gathers are in flight together (1: the two-stage pipeline of the launches that fill
// the GPU, at GTO_OBS_MIN_WAVES waves per SIMD; 8: the launches with few instances in flight, two waves per SIMD)
#ifndef GTO_OBS_MAIN_PD
#define GTO_OBS_MAIN_PD 1
#endif
// SWEEP: the launch behind an itemized launch that was laid out over an ESTIMATE of the item list's length (gto_api.hip,
// launch_obstacle): a small crew of workgroups that walks the items from `item0` on, in steps of the crew -- none, when the
// estimate held.  Nine tenths of the workgroups of a launch laid out over every (job, group) pair found no item and
// left; starting them costs the other lanes' launches dispatch slots (+7 % trajectories/s without them).  The loop lives
// in a variant of its own because a back edge around this kernel's body keeps every argument alive through it (200
// spilled SGPRs, +150 B of scratch): the crew pays that, the launch everybody runs through does not.
// HOT: the launches of the solve loop under the shipped gradient mode (fixed_mode == 0, voxel records with central
// differences): the value-only gather, the init pass's virtual waypoints and the static-link bookkeeping are compiled out of
// the variant that runs nine launches in ten, with the scalars and branches they kept alive in its loop.
// ROOM: the launch leaves BatchPtrs::room behind for every (waypoint, link) it looks at (emptiness certificates: the
// itemized launches of the rounds with few instances in flight, behind k_lm_step<8, 4>'s cert_tail)
template <int NP, int PD = GTO_OBS_MAIN_PD, bool SWEEP = false, bool HOT = false, bool ROOM = false>
__global__ __launch_bounds__(256, PD == GTO_OBS_MAIN_PD ? GTO_OBS_MIN_WAVES : 2) void k_obstacle_gram(const int32_t* __restrict__ jobs_par, const int32_t* __restrict__ njobs_par,
                                                       const int2* __restrict__ items_par, const int32_t* __restrict__ nitems_par, int nG_pre, uint32_t m_nG_pre,
                                                       int n_regular, int B,
                                                       const RobotDev* __restrict__ rb, const double* __restrict__ px,
                                                       const double* __restrict__ py, const double* __restrict__ pz,
                                                       const Chunk* __restrict__ chunks, const SceneDev* __restrict__ scenes,
                                                       BatchPtrs bp, SolveParams sp, int t_begin, int nT,
                                                       int fixed_mode_rt, ObsGeom geo, int item0) {
  const int fixed_mode = HOT ? 0 : fixed_mode_rt;
  // the crew's workgroups leave at once when the launch in front of them covered the list (nearly always): before the
  // kernel's entry code, whose spilled scalars alone are 370 instructions a wave
  if constexpr (SWEEP) {
    if ((int)blockIdx.x + item0 >= nitems_par[0]) return;
  }
  extern __shared__ __attribute__((aligned(16))) double smem_obs[];
  __shared__ int s_wcount[4];
  __shared__ unsigned s_touched[GTO_MAX_TG];  // per waypoint of the group: links whose Gram got a contribution
  // for the epilogue: per output entry (i, j) / (i) of the block the links both joints / the joint sit above
  // (RobotDev::entry_links) -- staged here, beside the kinematics' tables, so that the projection walks the set bits of
  // (touched & mask) in LDS instead of asking the robot table in memory link by link
  __shared__ unsigned s_emask[NP * NP + NP];
  // (ROOM) per (waypoint of the group, link): index steps the tightest culled sphere of the link is clear of the nearest
  // non-zero record, beyond its culling radius (-1: a sphere survives; 127: no sphere of the link was tested)
  int* s_room = nullptr;
  if constexpr (ROOM) {
    __shared__ int s_room_store[GTO_MAX_TG * GTO_MAX_LINKS];
    s_room = s_room_store;
  }

  const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = geo.L, n = geo.n, T = sp.T, F = geo.F, TG = geo.TG, cap_active = geo.cap_active;
  typedef Blk<NP> BK;
  const ObsLds& lay = geo.lay;
  double* s_vis = smem_obs + lay.vis;
  double* s_screw = smem_obs + lay.screw;
  double* s_gram = smem_obs + lay.gram;
  double* s_out = smem_obs + lay.out;
  double* s_list = smem_obs + lay.list;
  double* s_ktab = smem_obs + lay.uni;          // prologue only
  double* s_sc = s_ktab + fk_tab_doubles(F, L, n);  // prologue only
  int2* s_active = reinterpret_cast<int2*>(smem_obs + lay.active);  // (link | waypoint << 16, start | count << 16)
  double* s_u = s_list;    // [TG][L][NP][6] in the epilogue

  // The first sixteen dwords of the kernel's arguments arrive in registers with the wave (kernarg preload, -mllvm
  // -amdgpu-kernarg-preload-count=16): jobs_par / njobs_par (this round's job list and its length; null outside the solve
  // loop), the group count and its division magic.  The job's entry is therefore requested by the wave's first
  // instructions, beside the loads of the other arguments instead of behind them.  No clamp on the index: the host sizes
  // the launch by an upper bound of the list's length, rounded up to eight, and the list has that much slack behind it.
  const bool listed = jobs_par != nullptr;  // solve loop: workgroups are laid out over the job list
  // The goal-term workgroups come FIRST in the grid (a multiple of eight of them, so that block % 8 of the others is
  // unchanged): the dispatcher starts blocks in order, a goal job is one wavefront's serial work of 15-27 us, and behind
  // the regular blocks it was the tail of every launch (a launch whose regular blocks return at once still took 28 us).
  const int n_goal_wg = gridDim.x - n_regular;
  const bool is_goal = bid < n_goal_wg;
  int rbid = bid - n_goal_wg + (SWEEP ? item0 : 0);
  const int xcd = rbid & 7, kb = rbid >> 3;
  // (group-major block order with the groups near the goal, whose gather loops are the long ones, first: a launch alone on
  // the GPU gets 5 % shorter, the rate with four lanes in flight does not move: measured, not kept)
  const int nG = nG_pre;
  const int kbq = fast_div(kb, nG, m_nG_pre);
  int bi = kbq * 8 + xcd, grp_id = kb - kbq * nG;
  const int gi_ = bid * 4 + wave;  // goal workgroups: job of this wavefront
  // Rounds behind a step kernel that ran the broad phase itself (items_par != null; prebroad_tail): the regular workgroups
  // are laid out over the list of (job, group) pairs that have to be looked at; the others were settled without a workgroup.
  const bool itemized = items_par != nullptr && !is_goal;
  int n_act = B, le = is_goal ? gi_ : bi;
#ifdef GTO_DEBUG_LONGEST_WG
  int grp_id_flag = 0;
#endif
  if (listed) {  // one round trip for the list's length and the entry, requested by the wave's first instructions
    if (itemized) {
      n_act = nitems_par[0];
      const int2 it = items_par[rbid];
      le = it.x;
      bi = rbid < n_act ? it.y >> 8 : n_act;  // the job (its position in the job list: the joint values are stored by it)
      grp_id = it.y & 0x7f;
#ifdef GTO_DEBUG_LONGEST_WG
      grp_id_flag = it.y;
#endif
    } else {
      n_act = njobs_par[0];
      le = jobs_par[le];
    }
  }
  const int par = sp.parity, NS = sp.kcap + 1;
  // blockIdx -> (job, waypoint group), bijective, with job % 8 == blockIdx % 8; a job is one candidate trial trajectory of
  // one instance (outside the solve loop: the instance itself)
  // (Measured and not kept: a fixed crew of workgroups walking the items of a launch that fills the GPU, drawing them from
  // a counter per XCD or dealing them round-robin: 150 k and 218 k trajectories/s against 257 k.  The heavy items, 4 % with
  // more than 60 surviving chunks, decide when a launch ends, and the dispatcher balances them better than a static deal.)
  if (!SWEEP && listed && bid == 0 && tid == 0) {  // the lists this round's step kernel fills
    bp.nlive[GTO_NLIVE(1 - par)] = bp.nlive[GTO_NJOBS(1 - par)] = 0;
    bp.nlive[8 + 1 - par] = 0;
  }
  if (is_goal) {
    // Extra workgroups: goal-set terms and velocity term of the candidate trajectories.  The step
    // kernel only needs them at its NEXT launch, so they ride in the shadow of the obstacle evaluation instead of sitting
    // on the serial path between two launches (35 K cycles a job against 42 K for the heaviest regular workgroup of a
    // lone instance).  Four jobs per workgroup, one per wavefront: a goal workgroup holds a whole workgroup's LDS and wave
    // slots for one long serial job, so packing four of them into one costs a quarter of the slot time (a workgroup per
    // job with the kinematics on the matrix cores takes 22 K cycles, and 11 % off the saturated rate: measured, not kept)
    if (gi_ >= n_act || le < 0) return;
    const int bg = le & GTO_JOB_MASK, gcand = le >> GTO_JOB_SHIFT;
    // a fresh instance evaluates its seed, whose goal terms k_lm_init already produced
    if (bp.state[bg].done || (listed && bp.state[bg].first)) return;
    double* s_fr2 = smem_obs + wave * GTO_GOAL_SCRATCH(NP);  // [2][GTO_MAX_FRAMES*12]
    double* s_q2 = s_fr2 + 2 * GTO_MAX_FRAMES * 12;      // [2][GTO_MAX_DOF]
    double* s_ga = s_q2 + 2 * GTO_MAX_DOF;               // [48]
    double* s_gs = s_ga + 48;                            // [2][NP*6]
#ifdef GTO_DEBUG_LONGEST_WG
    const bool dbg_goal = bp.dbg && bg == 0 && gcand == 0 && lane == 0;
    if (dbg_goal) bp.dbg[31] = clock64();
#endif
    trial_goal_terms_wave<NP>(rb, bp, sp, B, bg, lane, listed ? GTO_JOB_SET(le) : (bp.state[bg].slot + 1 + gcand) % NS, gcand, bp.state + bg, s_q2, s_fr2, s_ga, s_gs);
#ifdef GTO_DEBUG_LONGEST_WG
    if (dbg_goal) bp.dbg[32] = clock64();
#endif
    return;
  }
  // (every exit below is taken by the whole workgroup; in the crew variant it leads to the next item)
  for (;;) {
  {
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
  const int ng = static_only ? 1 : (inter ? geo.q_nT + (grp_id < geo.r_nT ? 1 : 0) : min(TG, (fixed_mode ? 2 : t_begin + nT) - t0v));  // waypoints in this group
  auto wp = [&](int kq_) { return w0 + kq_ * wstep; };    // waypoint of the group's kq-th member
  if (sp.dbg_cut == 5 && (le >= 0 || bi < 0)) goto next_item;  // (timing experiments: the scalar part of the entry alone)

  // ---- prologue.  Every global load the kinematics need is issued here, before the first branch that depends on one
  // of them: ONE memory round trip for the job-list entry, the joint values of the frames (written by the step kernel,
  // indexed by job: no dependence on the instance id) and the operand table of fk_mfma_tree.  The index is clamped: the
  // host sizes the launch by an upper bound of the list's length.
  const int bic = min(bi, (listed ? bp.cap * sp.kcap : B) - 1);
  // (F: frames of the compact tree, gto_device.h; the joint values are stored by the robot's own frames, Fq of them)
  const int Fq = geo.Fq;
  const double* __restrict__ qfp = listed ? bp.qfs + ((((size_t)par * bp.cap * sp.kcap + bic) * T) + w0) * Fq : bp.qf + ((size_t)bic * T + w0) * Fq;
  const int qstride = wstep * Fq;  // offset per group member
  const int tq = fast_div(tid, F, geo.m_F);  // member of the group this thread's frame belongs to
  const int fo_t = tid < ng * F ? rb->cf_orig[tid - tq * F] : -1;  // the frame whose joint value this compact frame takes (-1: world)
  const int jtv = tid < ng * F ? rb->cf_type[tid - tq * F] : GTO_JOINT_FIXED;
  const double qfv = fo_t >= 0 ? qfp[fo_t + tq * qstride] : 0.0;
  const int nt = geo.nt;  // rb->fk_tab_c is packed for exactly this (F, L, n)
  double tabv[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) tabv[u] = tid + 256 * u < nt ? rb->fk_tab_c[tid + 256 * u] : 0.0;
  if ((itemized ? rbid : bi) >= n_act || le < 0) goto next_item;
  const int b = le & GTO_JOB_MASK, cand = le >> GTO_JOB_SHIFT;
  // inside the solve loop the evaluation outputs are sparse (BatchPtrs::wrec): a record per waypoint, no zero blocks
  const bool sparse = listed;
  const InstState* st = bp.state + b;
  const bool dbg_wg = bp.dbg && b == 0 && grp_id == nG - 1;
#ifdef GTO_DEBUG_LONGEST_WG
  const long long t_wg0 = bp.dbg ? clock64() : 0;
#endif
  if (dbg_wg && tid == 0) bp.dbg[10] = clock64();
  if (sp.dbg_cut == 6) goto next_item;
  // the lists never hold a finished instance, and a list entry names the block set to write: nothing of the instance's
  // state is read on this path (a dependent memory round trip in front of the kinematics otherwise)
  if (!listed && st->done) goto next_item;
  const int set_out = listed ? GTO_JOB_SET(le) : (st->slot + 1 + cand) % NS;
  // Everything the broad phase needs that does not depend on the kinematics is requested here, three dependent round
  // trips (scene index -> scene descriptor, chunk descriptor) that would otherwise follow the kinematics one by one
  const int C = geo.C;
  double ssfix = 0.0;  // constant contribution of the static links (measured once at init), added in the epilogue
  if (!fixed_mode && tid < ng) ssfix = bp.ss_fixed[4 * b + (wp(tid) < sp.ts ? 2 : 3)];
#pragma unroll
  for (int u = 0; u < 6; ++u)
    if (tid + 256 * u < nt) s_ktab[tid + 256 * u] = tabv[u];
  for (int k = tid + 256 * 6; k < nt; k += 256) s_ktab[k] = rb->fk_tab_c[k];  // very large robots
  if (sp.dbg_cut == 7) goto next_item;
  // sin/cos of every (waypoint, joint), one lane each
  if (tid < ng * F) {
    double a = 0.0, c = 1.0;
    if (jtv == GTO_JOINT_REVOLUTE) sincos(qfv, &a, &c);
    else if (jtv == GTO_JOINT_PRISMATIC) a = qfv;
    s_sc[2 * tid] = a;
    s_sc[2 * tid + 1] = c;
  }
  for (int idx = tid + 256; idx < ng * F; idx += 256) {  // waypoint groups of very large robots
    const int fo_ = rb->cf_orig[idx % F], jt = rb->cf_type[idx % F];
    const double qv = fo_ >= 0 ? qfp[fo_ + (idx / F) * qstride] : 0.0;
    double a = 0.0, c = 1.0;
    if (jt == GTO_JOINT_REVOLUTE) sincos(qv, &a, &c);
    else if (jt == GTO_JOINT_PRISMATIC) a = qv;
    s_sc[2 * idx] = a;
    s_sc[2 * idx + 1] = c;
  }
  if (tid < GTO_MAX_TG) s_touched[tid] = 0u;
  if constexpr (ROOM) {
    for (int i = tid; i < GTO_MAX_TG * GTO_MAX_LINKS; i += 256) s_room[i] = 127;
  }
  for (int e = tid; e < NP * NP + NP; e += 256) s_emask[e] = rb->entry_links[NP == 8 ? 0 : 1][e];  // (read in the epilogue, many barriers from here)
  __syncthreads();
  if (dbg_wg && tid == 0) bp.dbg[16] = clock64();
  if (sp.dbg_cut == 8) goto next_item;
  {
    double* s_X = s_sc + 2 * ng * F;
    if constexpr (PD == 1)
      fk_mfma_tree_compact(rb, s_ktab, ng, s_sc, s_X, reinterpret_cast<int*>(s_X + ng * 32 * F + 64), tid, s_vis, s_screw,
                           dbg_wg ? bp.dbg + 20 : nullptr, NP, F, geo.rounds);
    else
      fk_mfma_tree_batched(rb, s_ktab, ng, s_sc, s_X, reinterpret_cast<int*>(s_X + ng * 32 * F + 64), tid, s_vis, s_screw,
                           dbg_wg ? bp.dbg + 20 : nullptr, NP, F, geo.rounds);
  }
  __syncthreads();
  if (dbg_wg && tid == 0) bp.dbg[11] = clock64();
  if (sp.dbg_cut == 1) goto next_item;

  const SceneDev sc = scenes[bp.scene_id[b]];
  const double bx = bp.base_pos[3 * b], by = bp.base_pos[3 * b + 1], bz = bp.base_pos[3 * b + 2];
  const double cx = (bx - sc.ox) * sc.rinv, cy = (by - sc.oy) * sc.rinv, cz = (bz - sc.oz) * sc.rinv;
  const bool need_grad = HOT ? true : (!fixed_mode && sp.grad_mode == GTO_GRAD_CENTRAL_DIFF);
  const int nz = sc.nz;

  // ---- broad phase: one thread per (waypoint, chunk) transforms the chunk's bounding-sphere centre and
  // looks up the Chebyshev distance to the nearest non-zero voxel; a chunk whose sphere cannot reach one (R index steps
  // per axis at most: gto_device.h, GTO_BROAD_MARGIN) contributes exact zeros and is skipped.  Survivors keep (waypoint, link) order (ballot prefix): a wave still sees few key changes.
  auto use_all = [&](int kq_) { return static_only ? (t0v == 2) : (wp(kq_) < sp.ts); };
#ifdef GTO_DEBUG_LONGEST_WG
  __shared__ int s_dbg_room;  // least index shift any culled chunk of the group tolerates before it could survive
  if (tid == 0) s_dbg_room = 127;
  __syncthreads();
#endif
  int na_run = 0;
  for (int base_c = 0; base_c < ng * C; base_c += 256) {
    const int gi = base_c + tid;
    bool keep = false;
    int2 desc2 = make_int2(0, 0);
    if (gi < ng * C) {
      const int kq = fast_div(gi, C, geo.m_C), ci = gi - kq * C;
      const Chunk cc = chunks[ci];
      desc2 = make_int2(cc.link | (kq << 16), cc.start | (cc.count << 16));
      const bool is_static = cc.pad != 0;  // link not moved by any optimised joint
      const double* V = s_vis + (kq * L + cc.link) * 12;
      const double u0 = (V[0] * cc.cx + V[1] * cc.cy + V[2] * cc.cz + V[3] + bx - sc.ox) * sc.rinv;
      const double u1 = (V[4] * cc.cx + V[5] * cc.cy + V[6] * cc.cz + V[7] + by - sc.oy) * sc.rinv;
      const double u2 = (V[8] * cc.cx + V[9] * cc.cy + V[10] * cc.cz + V[11] + bz - sc.oz) * sc.rinv;
      const int R = (int)ceil(cc.r * sc.rinv + 1e-6) + GTO_BROAD_MARGIN;
      const int k0 = (int)floor(u0), k1 = (int)floor(u1), k2 = (int)floor(u2);
      keep = true;
      if ((is_static && !fixed_mode) || (!is_static && static_only)) {
        keep = false;  // static links are accounted once at init; the static-only pass ignores the rest
      } else
      // Spheres that stick out of the grid are tested at the clipped voxel of their centre: the gather clips a point's
      // indices the same way, and clipping moves two indices no further apart, so the points' voxels are still within R
      // of that one.  (Until round 3 only spheres wholly inside the grid could be culled.)
      if (R < GTO_DIST_CAP) {
        const int c0 = min(max(k0, 0), sc.nx - 1), c1 = min(max(k1, 0), sc.ny - 1), c2 = min(max(k2, 0), sc.nz - 1);
        const uint8_t* __restrict__ dist = use_all(kq) ? sc.d_all : sc.d_obs;
        const int dd = (int)as_global(dist)[c2 + nz * (c1 + sc.ny * c0)];
        keep = dd <= R;
        if constexpr (ROOM) {
          if (!keep) atomicMin(&s_room[kq * L + cc.link], dd - R - 1);
        }
#ifdef GTO_DEBUG_LONGEST_WG
        if (!keep) atomicMin(&s_dbg_room, dd - R - 1);
#endif
      }
      if constexpr (ROOM) {
        if (keep) atomicMin(&s_room[kq * L + cc.link], -1);
      }
#ifdef GTO_DEBUG_LONGEST_WG
      if (keep) atomicMin(&s_dbg_room, -1);
#endif
    }
    const unsigned long long bm = __ballot(keep);
    if (lane == 0) s_wcount[wave] = __popcll(bm);
    __syncthreads();
    // every thread keeps the running count itself (the same four numbers for all): two barriers per pass, not three
    const int w0c = s_wcount[0], w1c = s_wcount[1], w2c = s_wcount[2], w3c = s_wcount[3];
    const int woff = na_run + (wave > 0 ? w0c : 0) + (wave > 1 ? w1c : 0) + (wave > 2 ? w2c : 0);
    if (keep) {
      const int pos = woff + __popcll(bm & ((1ull << lane) - 1ull));
      if (pos < cap_active) s_active[pos] = desc2;
    }
    na_run = min(na_run + w0c + w1c + w2c + w3c, cap_active);
    __syncthreads();
  }
  const int NA = na_run;
  if constexpr (ROOM) {
    // what this look found, in metres (cert_tail subtracts what a candidate's step can move a point of the link by): a
    // sphere whose centre's voxel is d > R index steps from the nearest non-zero record stays culled while its points move
    // by at most (d - R - 1) voxels -- they then lie within R + (d - R - 1) < d index steps of that voxel.  The last barrier
    // of the loop above is behind every atomicMin.
    if (bp.room && !fixed_mode && tid < ng * L) {
      const int kq = fast_div(tid, L, geo.m_L), l = tid - kq * L, v = s_room[tid];
      bp.room[(((size_t)set_out * B + b) * T + wp(kq)) * L + l] = v == 127 ? 1e30f : (v < 1 ? -1.f : (float)((double)v * sc.res * 0.999999));
    }
  }
#ifdef GTO_DEBUG_LONGEST_WG
  // verification of the step kernel's broad phase (GTO_DEBUG_CUT=10): a group it marked as settled is looked at anyway and
  // must come out without a contribution (counted at the end of the workgroup).  Without a surviving chunk, too, as long
  // as the step kernel tests the chunks' own spheres (GTO_PB_MERGE=1); a sphere over several chunks can prove what the
  // chunks' spheres, which stick out of it, cannot: both tests are sufficient, neither is necessary
  if (sp.dbg_cut == 10 && bp.dbg && itemized && (grp_id_flag & 0x80) && tid == 0) {
    atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 49), 1ull);
    if (NA != 0) atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 50), 1ull);
  }
#endif
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
  if (sp.dbg_cut == 2) goto next_item;
  // No chunk of the group can reach a non-zero voxel (58 % of the workgroups of the bench workload): the blocks are exact
  // zeros and the sum of c^2 is the static links' constant, written here without the fold, the projection and their
  // six barriers.  (0.0 + x: the bits of the long way round, where the constant is added to a sum of zeros.)
  if (NA == 0 && !fixed_mode) {
#ifdef GTO_DEBUG_LONGEST_WG
    if (bp.dbg && tid == 0) {
      atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 128 + max(0, min(s_dbg_room, 63))), 1ull);
      atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 64), 1ull);
      atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 192), (unsigned long long)(clock64() - t_wg0));
    }
#endif
    if (sparse) {  // the sum of c^2 and a cleared flag per waypoint: no zero block is written (nor read back)
      const double ssf_ = __shfl(ssfix, lane >> 3, 64);  // (ssfix lives in lane kq of wave 0)
      if (tid < ng * 8) bp.wrec[(((size_t)set_out * B + b) * T + wp(tid >> 3)) * 8 + (tid & 7)] = (tid & 7) == 0 ? 0.0 + ssf_ : 0.0;
      goto next_item;
    }
    double* out = bp.blocks + (((size_t)set_out * B + b) * T + w0) * BK::STRIDE;
    const int ostride = (wstep - 1) * BK::STRIDE;
    for (int i = tid; i < ng * BK::STRIDE; i += 256) {
      const int kq = i / BK::STRIDE;
      if (i - kq * BK::STRIDE != BK::SS) out[i + kq * ostride] = 0.0;
    }
    if (tid < ng) out[tid * (BK::STRIDE + ostride) + BK::SS] = 0.0 + ssfix;
    goto next_item;
  }
  // the kinematics scratch is dead: its region now holds the Gram accumulators
  for (int i = tid; i < ng * L * GTO_GRAM; i += 256) s_gram[i] = 0.0;
  __syncthreads();

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
  // packed Gram index of D entry (row, col), row <= col < 7: 21 wrench-Gram entries, then c * wrench (6), then c^2 (slot 27:
  // the sum of c^2 over ALL the key's points -- in gradient mode every point with a non-zero cost is on the list, with a zero
  // wrench if its gradient vanishes, so the corner of the fold IS that sum and no reduction across the wave is needed)
  auto gram_index = [](int row, int col) { return col < 6 ? sym6(row, col) : (row < 6 ? 21 + row : 27); };
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
    /* a key is flushed once, by the one wave that folds it, onto the zeros the accumulators were cleared to: plain stores */ \
    if (w0_) gdst_[gk0] = gD[0];                                                             \
    if (w1_) gdst_[gk1] = gD[1];                                                             \
    if (__ballot(w0_ || (w1_ && gk1 != 27)) && lane == 0) atomicOr(&s_touched[fk_], 1u << fl_); \
    gD = gto_v4f64{0.0, 0.0, 0.0, 0.0};                                                      \
    /* sum of c^2 of the key (slot 27): per key, summed over the links in link order in the epilogue, so the value does  \
       not depend on how the chunks were dealt to the waves (nor on the group size).  Gradient mode: the fold's corner,  \
       above; value-only mode keeps no list: the lanes' partial sums, reduced across the wave */ \
    if (!need_grad) {                                                                        \
      const double sw_ = wave_sum(ss);                                                       \
      if (lane == 0) gdst_[27] = sw_;                                                        \
      ss = 0.0;                                                                              \
    }                                                                                        \
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
    // (three chained multiply-adds per coordinate; the voxel index does not depend on their rounding: fallback below)
    const double y0 = fma(V[0], x0, fma(V[1], x1, fma(V[2], x2, V[3])));
    const double y1 = fma(V[4], x0, fma(V[5], x1, fma(V[6], x2, V[7])));
    const double y2 = fma(V[8], x0, fma(V[9], x1, fma(V[10], x2, V[11])));
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
    const int ix = clamp_index((int)k0, sc.nx - 1);  // v_cvt_i32_f64 saturates, NaN -> 0
    const int iy = clamp_index((int)k1, sc.ny - 1);
    const int iz = clamp_index((int)k2, sc.nz - 1);
    // 24-bit multiply-adds (full rate; the 32-bit ones run at a quarter of it): nx ny <= 2^24 and nz < 2^24 are checked when
    // the scene is set
    const unsigned off = __umul24((unsigned)nz, __umul24((unsigned)sc.ny, (unsigned)ix) + (unsigned)iy) + (unsigned)iz;
    st_.key = ch.key;
    st_.count = ch.count;
    st_.y0 = y0, st_.y1 = y1, st_.y2 = y2;
    st_.fval = 0.f;
    st_.rec = make_double4(0.0, 0.0, 0.0, 0.0);
    if (!need_grad) {
      st_.fval = as_global(pre ? sc.c_all : sc.c_obs)[off];
    } else {
      // one 32-B voxel record: cost + central differences (gto/sdf_callback.py:90-114); the divisor
      // stays 2*res also at clipped borders.  Two 16-B loads, one cache line.
      st_.rec = load_record(pre ? sc.r_all : sc.r_obs, off);
    }
  };
  // stage B: consume a chunk whose record (or field value) has arrived (a macro, expanded in both loop variants: the tuned
  // variant's register allocation does not survive a lambda here)
#define GTO_CONSUME(cur)                                                                                              \
  do {                                                                                                                \
      if (cur.key != cur_key) {                                                                                     \
        if (cur_key >= 0) {                                                                                         \
          GTO_FLUSH(cur_key);                                                                                       \
        }                                                                                                           \
        cur_key = cur.key;                                                                                          \
      }                                                                                                             \
      n_gathered += cur.count;                                                                                      \
      const bool valid = lane < cur.count;                                                                          \
      if (!need_grad) {                                                                                             \
        const double cval = valid ? (double)cur.fval : 0.0;                                                         \
        ss = fma(cval, cval, ss);                                                                                   \
      } else {                                                                                                      \
        const double4 lo4 = cur.rec;                                                                                \
        const double y0 = cur.y0, y1 = cur.y1, y2 = cur.y2;                                                         \
        const double cval = (double)__builtin_bit_cast(float, (unsigned)__double2loint(lo4.w));                      \
        const double w0 = lo4.x * sc.inv2r;                                                                         \
        const double w1 = lo4.y * sc.inv2r;                                                                         \
        const double w2 = lo4.z * sc.inv2r;                                                                         \
        /* on the list: every point of the chunk with a non-zero gradient or a non-zero cost (its c^2 is summed by the fold) */ \
        const bool act = valid && (w0 != 0.0 || w1 != 0.0 || w2 != 0.0 || cval != 0.0);                             \
        const unsigned long long am = __ballot(act);                                                                \
        if (am) {                                                                                                   \
          if (act) {                                                                                                \
            double2* e = reinterpret_cast<double2*>(lst + (cnt + __popcll(am & ((1ull << lane) - 1ull))) * 8);      \
            e[0] = make_double2(y1 * w2 - y2 * w1, y2 * w0 - y0 * w2);                                              \
            e[1] = make_double2(y0 * w1 - y1 * w0, w0);                                                             \
            e[2] = make_double2(w1, w2);                                                                            \
            reinterpret_cast<double*>(e)[6] = cval;                                                                 \
          }                                                                                                         \
          cnt += __popcll(am);                                                                                      \
          if (cnt > GTO_LIST_CAP - 64) GTO_DRAIN();                                                                 \
        }                                                                                                           \
      }                                                                                                             \
                                                                                                                    \
  } while (0)
  if constexpr (PD == 1) {
    // (Measured and not kept, round 5: the loop written out for two staging slots used alternately, so that no register
    // copy "current = next" ends an iteration -- the compiler still waits for the whole memory queue at the loop head,
    // because at 96 VGPRs it reuses the destination registers of loads in flight; and variants with two or four chunks'
    // gathers in flight at four / three waves per SIMD: 68.3 / 60.5 / 50.5 k against 68.5 k trajectories/s on the shelf
    // workload.  The loop is bound by instruction issue, not by the latency of its gathers.)
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
      GTO_CONSUME(cur);
      cur = nxt;
    }
  } else {
    // Launches with few instances in flight: the heaviest workgroup decides the round (a wave with the seven chunks of one
    // link next to an obstacle pays seven dependent trips to the records, 3 K cycles each), registers are plentiful (two
    // waves per SIMD).  Batches of up to PD chunks: all their point coordinates are requested, then all their records,
    // then they are consumed in the same order as ever (same folds, same bits).
#pragma unroll 1
    for (int cb = c0; cb < c1; cb += PD) {
      ChunkLite ch[PD];
      double x0[PD], x1[PD], x2[PD];
      Staged stg[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i)
        if (cb + i < c1) load_chunk(cb + i, ch[i], x0[i], x1[i], x2[i]);  // wave-uniform
#pragma unroll
      for (int i = 0; i < PD; ++i)
        if (cb + i < c1) stage_a(ch[i], x0[i], x1[i], x2[i], stg[i]);
#pragma unroll
      for (int i = 0; i < PD; ++i)
        if (cb + i < c1) GTO_CONSUME(stg[i]);
    }
  }
  if (cur_key >= 0) {
    GTO_FLUSH(cur_key);
  }
#undef GTO_CONSUME
#undef GTO_FLUSH
#undef GTO_DRAIN
  // work counter of a profiled solve (bench.py prices the roofline on it): 64 cells so that the few thousand waves of a
  // launch that gathered anything do not queue up on one address
  if (bp.work && !fixed_mode && lane == 0 && n_gathered) atomicAdd(bp.work + (bid & 63), (unsigned long long)n_gathered);
  if (dbg_wg && tid == 0) bp.dbg[13] = clock64();
  if (sp.dbg_cut == 3) goto next_item;
  __syncthreads();
  // Chunks survived the broad phase but no point of theirs has a non-zero gradient (37 % of the workgroups of the bench
  // workload; with the 58 % above, 95 % of all): zero blocks again, the sum of c^2 from the keys in link order as below
  if (!fixed_mode) {
    unsigned any_ = 0u;
    for (int q = 0; q < ng; ++q) any_ |= s_touched[q];
    if (!any_) {
      if (sparse) {
        const double ssf_ = __shfl(ssfix, lane >> 3, 64);
        if (tid < ng * 8) {  // eight lanes per waypoint: the whole record in one store
          const int kq = tid >> 3;
          double v = 0.0;
          for (int l = 0; l < L; ++l) v += s_gram[(kq * L + l) * GTO_GRAM + 27];
          bp.wrec[(((size_t)set_out * B + b) * T + wp(kq)) * 8 + (tid & 7)] = (tid & 7) == 0 ? v + ssf_ : 0.0;
        }
        goto next_item;
      }
      double* out = bp.blocks + (((size_t)set_out * B + b) * T + w0) * BK::STRIDE;
      const int ostride = (wstep - 1) * BK::STRIDE;
      for (int i = tid; i < ng * BK::STRIDE; i += 256) {
        const int kq = i / BK::STRIDE;
        if (i - kq * BK::STRIDE != BK::SS) out[i + kq * ostride] = 0.0;
      }
      if (tid < ng) {
        double v = 0.0;
        for (int l = 0; l < L; ++l) v += s_gram[(tid * L + l) * GTO_GRAM + 27];
        out[tid * (BK::STRIDE + ostride) + BK::SS] = v + ssfix;
      }
      goto next_item;
    }
  }
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
      const bool on = (s_emask[NP * NP + j] >> l) & 1u;
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
      unsigned m_ = touched & s_emask[e];  // links that contribute to this entry, walked in link order
      if (e < NP * NP) {
        const int i = e / NP, j = e % NP;
        double v = 0.0;
        const double* si = screw + 6 * i;
        while (m_) {
          const int l = __ffs(m_) - 1;
          m_ &= m_ - 1u;
          const double* u = s_u + ((kq * L + l) * NP + j) * 6;
          v += si[0] * u[0] + si[1] * u[1] + si[2] * u[2] + si[3] * u[3] + si[4] * u[4] + si[5] * u[5];
        }
        s_out[kq * BK::STRIDE + BK::JTJ + e] = v;
      } else {
        const int i = e - NP * NP;
        double v = 0.0;
        const double* si = screw + 6 * i;
        while (m_) {
          const int l = __ffs(m_) - 1;
          m_ &= m_ - 1u;
          const double* vv = s_gram + (kq * L + l) * GTO_GRAM + 21;
          v += si[0] * vv[0] + si[1] * vv[1] + si[2] * vv[2] + si[3] * vv[3] + si[4] * vv[4] + si[5] * vv[5];
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
#ifdef GTO_DEBUG_LONGEST_WG  // (a compile-time switch: the extra live value costs the tuned variant registers)
  if (bp.dbg && tid == 0 && !fixed_mode) {  // the longest regular workgroup of the call: cycles << 16 | surviving chunks
    const unsigned long long v_ = ((unsigned long long)(clock64() - t_wg0) << 16) | (unsigned long long)min(NA, 65535);
    atomicMax(reinterpret_cast<unsigned long long*>(bp.dbg + 40), v_);
    // histogram of the surviving chunks per workgroup (cells 64..127), workgroups none of whose keys got a contribution
    atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 64 + min(NA, 63)), 1ull);
    atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 192 + min(NA, 63)), (unsigned long long)(clock64() - t_wg0));  // ticks they ran, summed (cells 192..255)
    unsigned any_ = 0;
    for (int q_ = 0; q_ < ng; ++q_) any_ |= s_touched[q_];
    if (!any_) atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 48), 1ull);
    // verification of the step kernel's broad phase: a group it settled must not get a contribution (cell 55: has to stay 0)
    if (sp.dbg_cut == 10 && itemized && (grp_id_flag & 0x80) && any_) atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 55), 1ull);
  }
#endif
  if (fixed_mode) {
    if (tid < ng) bp.ss_fixed[4 * b + t0v + tid] = s_out[tid * BK::STRIDE + BK::SS];
  } else {
    // add the constant contribution of the static links (measured once at init)
    if (tid < ng) s_out[tid * BK::STRIDE + BK::SS] += ssfix;
    __syncthreads();
    double* out = bp.blocks + (((size_t)set_out * B + b) * T + w0) * BK::STRIDE;
    const int ostride = (wstep - 1) * BK::STRIDE;
    if (sparse) {  // the blocks of the waypoints with a touched key, every waypoint's sum of c^2 and flag
      for (int i = tid; i < ng * BK::STRIDE; i += 256) {
        const int kq = i / BK::STRIDE;
        if (s_touched[kq]) out[i + kq * ostride] = s_out[i];
      }
      if (tid < ng * 8) {
        const int kq = tid >> 3, e = tid & 7;
        bp.wrec[(((size_t)set_out * B + b) * T + wp(kq)) * 8 + e] = e == 0 ? s_out[kq * BK::STRIDE + BK::SS] : (e == 1 && s_touched[kq] ? 1.0 : 0.0);
      }
    } else {
      for (int i = tid; i < ng * BK::STRIDE; i += 256) out[i + (i / BK::STRIDE) * ostride] = s_out[i];
    }
  }
  }
  next_item:
    if constexpr (!SWEEP) return;
    else {
      rbid += n_regular;
      if (!itemized || rbid >= n_act) return;
      __syncthreads();  // the item's LDS is free again
      const int2 it = items_par[rbid];
      le = it.x;
      bi = it.y >> 8;
      grp_id = it.y & 0x7f;
#ifdef GTO_DEBUG_LONGEST_WG
      grp_id_flag = it.y;
#endif
    }
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
#pragma unroll
  for (int r = 0; r < 3; ++r) {  // invt (optas/spatialmath.py:271-280)
#pragma unroll
    for (int c = 0; c < 3; ++c) inv[4 * r + c] = Te[4 * c + r];
    inv[4 * r + 3] = -(Te[r] * Te[3] + Te[4 + r] * Te[7] + Te[8 + r] * Te[11]);
  }
  aff_mul(inv, Tg, G);
#pragma unroll
  for (int i = 0; i < 12; ++i) RT[i] = RT16[i];
  if (S16) {
    double S[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) S[i] = S16[i];
    aff_mul(RT, S, RT);
  }
  aff_mul(RT, G, Y);
}

// sum_k || A p_k - Y p_k ||^2 = tr(D M D^T) + 2 d.(D mu) + K |d|^2, D = R_A - R_Y, d = t_A - t_Y
__device__ __forceinline__ double goal_cost_moments(const RobotDev* rb, const double* A, const double* Y) {
  double D[9], d[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) D[3 * r + c] = A[4 * r + c] - Y[4 * r + c];
    d[r] = A[4 * r + 3] - Y[4 * r + 3];
  }
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double dm[3];
#pragma unroll
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
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      R[3 * r + c] = A[4 * r + c];
      D[3 * r + c] = A[4 * r + c] - Y[4 * r + c];
    }
    t[r] = A[4 * r + 3];
    d[r] = A[4 * r + 3] - Y[4 * r + 3];
  }
  double Rmu[3], Dmu[3], RM[9], XX[9], N[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Rmu[r] = R[3 * r] * rb->grip_mu[0] + R[3 * r + 1] * rb->grip_mu[1] + R[3 * r + 2] * rb->grip_mu[2];
    Dmu[r] = D[3 * r] * rb->grip_mu[0] + D[3 * r + 1] * rb->grip_mu[1] + D[3 * r + 2] * rb->grip_mu[2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      RM[3 * r + c] = R[3 * r] * rb->grip_M[c] + R[3 * r + 1] * rb->grip_M[3 + c] + R[3 * r + 2] * rb->grip_M[6 + c];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // sum x x^T = R M R^T + (R mu) t^T + t (R mu)^T + K t t^T ;  N = R M D^T
      XX[3 * r + c] = RM[3 * r] * R[3 * c] + RM[3 * r + 1] * R[3 * c + 1] + RM[3 * r + 2] * R[3 * c + 2] +
                      Rmu[r] * t[c] + t[r] * Rmu[c] + K * t[r] * t[c];
      N[3 * r + c] = RM[3 * r] * D[3 * c] + RM[3 * r + 1] * D[3 * c + 1] + RM[3 * r + 2] * D[3 * c + 2];
    }
  const double trxx = XX[0] + XX[4] + XX[8];
  double sx[3] = {Rmu[0] + K * t[0], Rmu[1] + K * t[1], Rmu[2] + K * t[2]};
  double W[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) W[i] = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) W[6 * r + c] = ((r == c) ? trxx : 0.0) - XX[3 * r + c];
  // upper-right block [sum x]_x
  W[6 * 0 + 4] = -sx[2];
  W[6 * 0 + 5] = sx[1];
  W[6 * 1 + 3] = sx[2];
  W[6 * 1 + 5] = -sx[0];
  W[6 * 2 + 3] = -sx[1];
  W[6 * 2 + 4] = sx[0];
  W[6 * 3 + 3] = W[6 * 4 + 4] = W[6 * 5 + 5] = K;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
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
                                             int b, int lane, int trial, int cand, InstState* st, double* s_q, double* s_fr,
                                             double* s_gaff, double* s_gscr) {
  const int T = sp.T, n = rb->n_opt, ndof = rb->ndof;
  const double* Q0b = bp.Q0 + (size_t)b * ndof * T;
  const double* Qt = bp.Qtry + ((size_t)cand * B + b) * n * T;  // candidate `cand`, whose blocks go to set `trial`
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
    st->fgoal_try[cand] = go.f_goal;
    st->fvel_try[cand] = sp.w_vel * fv;
    st->argmin_try[cand] = go.argmin;
  }
}

// ------------------------------------------------------------------------------------------------
// Broad phase ahead of the obstacle launch (the rounds that fill the GPU).  Nine tenths of the obstacle
// kernel's workgroups of a table-top workload run entry -> kinematics -> broad phase only to learn that no bounding
// sphere of theirs can reach a non-zero voxel record.  The step kernel, which has just produced the trial trajectory, runs
// that test itself for all waypoints of the trajectory at once, with kinematics laid out for throughput instead of
// latency: one lane per (waypoint, row of the 3x4 transforms) walks the kinematic tree serially in registers (no matrix
// core, no LDS exchange, no barrier per level), the visual transforms go to LDS, then one lane per (waypoint, chunk)
// tests the chunk's bounding sphere exactly as k_obstacle_gram's broad phase does.  A waypoint group none of whose
// spheres survives is SETTLED here: its sums of c^2 (the static links' constant) and cleared block flags are written
// and it gets no workgroup of the obstacle kernel; the other groups go on the item list the next obstacle launch is laid
// out over.  Exact: a culled chunk contributes exact zeros.  The test is sound for any kinematics accurate to far less than
// the 1e-6 voxel of slack in the culling radius R = ceil(rho + 1e-6): a point of the chunk lies within rho (+ the
// difference between the two kinematics, ~1e-13) voxels of the sphere centre this function computes, so its voxel index
// is within R of that centre's on every axis, whichever of the two roundings of the centre is taken; the obstacle kernel,
// which recomputes the kinematics on the matrix cores for the groups it is given, may only find fewer survivors.
//   s_qt   [m][8]  trial joint values of the optimised joints (LDS)
//   s_work dead LDS of the step kernel: flags [64 ints] | sin, cos [m][F][2] | visual transforms [pw][L][12]
// Every thread of the workgroup must call it (barriers).  Returns nothing: outputs are ssw / nzb of the settled groups and
// the item list entries of the others.
// sphere tests a lane keeps in flight (their distance-field lookups): with spheres over runs of four chunks a lane of the
// reference's robots has three or four tests in all (4 / 5 / 8: 618.7 / 625.2 / 620.0 k trajectories/s on the default run)
#ifndef GTO_PB_BATCH
#define GTO_PB_BATCH 5
#endif
#ifndef GTO_DIAG_BATCH
#define GTO_DIAG_BATCH 4  // steps of the solve's diagonal stretch whose operands are fetched ahead of the dependent chain
#endif
#define GTO_PB_PARK 4  // frames whose transform a later, non-adjacent child needs (RobotDev::xst_slot): register slots
// bounding sphere of a chunk of a MOVING link, centre in the coordinates of the frame that carries the link (the link's visual
// origin applied on the host), with that frame (gto_create builds the table)
struct PbChunk {
  double cx, cy, cz, r;
  int32_t frame, pad;
};
// LDS of the tail.  The per-waypoint transforms, as FLOATS [pw][12 F + 4], sit at the start of the step kernel's dead
// region, in front of the trial point (which lives where the couplings s_e were: doubles [80 m, 88 m)); the tables (group
// flags, chunk spheres with their culling radius in voxels, frame constants, visual origins) go behind it if they fit there
// (up to the end of the kernel's LDS), otherwise in front of it at the price of fewer waypoints per pass.
// The tail is single-precision arithmetic throughout -- the joint angle is reduced to [-pi/4, pi/4] in double, everything
// after that (sin, cos, Rodrigues, the product with the frame's origin, the walk over the tree, the sphere centres) in
// floats: that is what lets the transforms of all waypoints of a 50-waypoint trajectory sit in LDS at once (one pass, one
// walk over the tree) and it takes the tail off the FP64 pipe, which issues at half the rate here (until round 5 the local
// transforms were formed in double and rounded on storage: phase A 6.5 K ticks of 28 K).  The price: every operation is off
// by at most 2^-24 of its result's size.  First order, per frame of the walk: an entry of the rotation part of a local
// transform carries at most 12 such errors (sin and cos 2 each, 1 - cos, three products and a sum, then the 3-term
// product with the origin and the origin's own rounding), i.e. 12 sqrt(3) 2^-24 |x| < 21 2^-24 |x| on a point x of that
// frame, its translation 5 2^-24 |t|; the product of the walk another 4 2^-24 (|x| + |t|); the centre and its voxel
// coordinate 8 2^-24 D more: less than 32 F 2^-24 D in all for a robot whose frame origins and surface points stay
// within D of each other.  The culling radius is widened by four times that (SolveParams::pb_eps, metres; gto_create), so
// the test stays sound: it keeps a few more spheres than the obstacle kernel's own, never fewer than the points need.
// Row of a frame in the step kernel's frame table (prebroad_tail, phase A: one lane per (waypoint, frame) reads its frame's
// sixteen floats): 20 floats apart, not 16 -- the sixteen-byte slots of eight consecutive frames then start on eight
// different groups of four LDS banks (with rows of 128 B of doubles, round 4, frames f and f + 2 started on the same bank and
// the lanes of a read were served six at a time: SQ_LDS_BANK_CONFLICT 4.4 cycles per LDS instruction of the kernel).
// GTO_PB_FT: doubles set aside per frame (the table shares the tail's region of doubles).
#define GTO_PB_FT 10
#define GTO_PB_FTF 20
__host__ __device__ inline int pb_table_doubles(int F, int Cm, int m, int npar) { return 32 + 4 * Cm + GTO_PB_FT * F + m * npar; }
__host__ __device__ inline size_t lm_lds_bytes(int T, int KL);  // (defined with the step kernel, below)
__host__ __device__ inline int pb_wp_stride(int F) { return 12 * F + 4; }  // floats; + 4: 16-byte rows, waypoints on different LDS banks
struct PbLayout {
  int tab0, pw;  // first double of the tables; waypoints per pass (0: the tail does not fit)
  __host__ __device__ PbLayout(int T, int F, int Cm, int npar) {
    const int m = T - 2, top = (int)(lm_lds_bytes(T, 1) / sizeof(double)), tabs = pb_table_doubles(F, Cm, m, npar);  // top: the one-candidate launch's LDS in doubles
    if (88 * m + tabs <= top) tab0 = 88 * m, pw = (160 * m) / pb_wp_stride(F);
    else tab0 = 80 * m - tabs, pw = tab0 > 0 ? (2 * tab0) / pb_wp_stride(F) : 0;
    if (pw > m) pw = m;
  }
};
// per-frame control word of the walk over the kinematic tree (RobotDev::pb_ctl, read through scalar loads):
//   bits 0-2  where the parent's row comes from: 0 identity (a root), 1 the previous frame, 2 + k parking slot k
//   bits 4-6  parking slot this frame's row goes to, plus one (0: none)
//   bit  8    a moving collision link hangs on this frame (its global transform is wanted)
#define GTO_PB_SRC(c) ((c)&7)
#define GTO_PB_PARKTO(c) (((c) >> 4) & 7)
#define GTO_PB_LINK(c) (((c) >> 8) & 1)

// sin and cos in single precision of an angle given in double: the reduction by pi/2 in double (any angle a joint can wind
// up to), the polynomials on [-pi/4, pi/4] in floats (errors below 2^-24: the next terms are r^11 / 11! and r^10 / 10!)
__device__ __forceinline__ void pb_sincosf(double x, float& sn, float& cs) {
  const double k = rint(x * 6.36619772367581382433e-01);
  const float r = (float)fma(-k, 6.07710050650619224932e-11, fma(-k, 1.57079632673412561417e+00, x));
  const float z = r * r;
  const float s0 = fmaf(r * z, fmaf(z, fmaf(z, fmaf(z, 2.7557319e-06f, -1.9841270e-04f), 8.3333333e-03f), -1.6666667e-01f), r);
  const float c0 = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.4801587e-05f, -1.3888889e-03f), 4.1666667e-02f), -0.5f), 1.f);
  const int q = (int)k & 3;
  sn = (q & 1) ? c0 : s0;
  cs = (q & 1) ? s0 : c0;
  if (q == 1 || q == 2) cs = -cs;
  if (q >= 2) sn = -sn;
}

template <int NT>
__device__ __forceinline__ void prebroad_tail(const RobotDev* __restrict__ rb, const PbChunk* __restrict__ pbc,
                                              const BatchPtrs& bp, const SolveParams& sp, int B, int b, int scene_b, int set_n, int pos_next,
                                              const double* __restrict__ s_qt, double* __restrict__ smem, int tid) {
  typedef float gto_f4 __attribute__((ext_vector_type(4)));
  const int lane = tid & 63;
  const int T = sp.T, m = T - 2, F = rb->n_frames, Cm = sp.pb_C, par = sp.parity;
  const int TG = sp.pb_tg, nG = sp.pb_ng, PW = sp.pb_pw, WS = pb_wp_stride(F);
  float* __restrict__ s_L = reinterpret_cast<float*>(smem);  // [PW][WS]: local transforms, then (frames with a moving link) global transforms
  double* __restrict__ s_tab = smem + sp.pb_tab0;
  int* __restrict__ s_flag = reinterpret_cast<int*>(s_tab);  // [64] per group: a sphere of it survives
  // chunk table, two arrays (one lane per chunk reads both: 16 and 4 bytes apart, no two lanes of a read on one bank; as
  // one array of 32-byte entries the frame numbers sat on four banks)
  gto_f4* __restrict__ s_ck = reinterpret_cast<gto_f4*>(s_tab + 32);  // [Cm]: centre and culling radius in voxels (floats)
  int* __restrict__ s_ckf = reinterpret_cast<int*>(s_tab + 32 + 2 * Cm);  // [Cm]: frame of the chunk's link
  float* __restrict__ s_ft = reinterpret_cast<float*>(s_tab + 32 + 4 * Cm);  // [F][GTO_PB_FTF]: origin (3x4), unit axis, (int) joint type | optimised joint + 1 << 4 | parameter slot + 1 << 9
  double* __restrict__ s_qp = s_tab + 32 + 4 * Cm + GTO_PB_FT * F;   // [m][npar]: values of the actuated joints that are not optimised
  const SceneDev sc = bp.scenes[__builtin_amdgcn_readfirstlane(scene_b)];
  const double bx = bp.base_pos[3 * b], by = bp.base_pos[3 * b + 1], bz = bp.base_pos[3 * b + 2];
  const bool dbg_t = bp.dbg && b == 0 && tid == 0;
  // ---- tables (one round trip)
  // the control words of the walk over the tree (phase B), all frames (at most GTO_MAX_FRAMES = 32) in one register, lane f
  // holding frame f's: the walk reads them by lane number (a scalar load per frame, issued a frame ahead, still cost its
  // wait in every iteration)
  const int ctlv = rb->pb_ctl[lane & (GTO_MAX_FRAMES - 1)];
  if (tid < 64) s_flag[tid] = 0;
  for (int c = tid; c < Cm; c += NT) {
    const PbChunk cc = pbc[c];
    gto_f4 v;
    v.x = (float)cc.cx, v.y = (float)cc.cy, v.z = (float)cc.cz;
    v.w = (float)((int)ceil((cc.r + sp.pb_eps) * sc.rinv + 1e-6) + GTO_BROAD_MARGIN);
    s_ck[c] = v;
    s_ckf[c] = cc.frame;
  }
  for (int i = tid; i < F * 16; i += NT) {
    const int f = i >> 4, e = i & 15;
    s_ft[GTO_PB_FTF * f + e] = e < 12 ? (float)rb->origin[f][e] : (e < 15 ? (float)rb->axis_unit[f][e - 12] : __int_as_float(rb->joint_type[f] | ((rb->opt_of_frame[f] + 1) << 4) | ((rb->pb_par[f] + 1) << 9)));
  }
  {  // joints that are actuated but not optimised: as k_lm_init left them (requested here, with the tables: one round trip)
    const double* __restrict__ qf0 = bp.qf + ((size_t)b * T + 2) * F;
    const int npar = sp.pb_npar;
    for (int i = tid; i < m * npar; i += NT) {
      const int sI = i / npar, k = i - sI * npar;
      s_qp[i] = qf0[(size_t)sI * F + rb->pb_parf[k]];
    }
  }
  if (dbg_t) bp.dbg[33] = clock64();
  __syncthreads();
  if (dbg_t) bp.dbg[34] = clock64();
  const int row = tid & 3;
  for (int p0 = 0; p0 < m; p0 += PW) {
    const int pwc = min(PW, m - p0);
    // ---- A: local transform L_f = O_f M_f(q) of every (waypoint of the pass, frame), one lane each: nothing serial
    for (int idx = tid; idx < pwc * F; idx += NT) {
      const int w = fast_div(idx, F, sp.pb_mF), f = idx - w * F;
      const gto_f4* __restrict__ ft = reinterpret_cast<const gto_f4*>(s_ft + GTO_PB_FTF * f);
      const gto_f4 o0 = ft[0], o1 = ft[1], o2 = ft[2], ux = ft[3];
      const int tj = __float_as_int(ux.w), jt = tj & 15, j = ((tj >> 4) & 31) - 1, kp = (tj >> 9) - 1;
      gto_f4* __restrict__ Lo = reinterpret_cast<gto_f4*>(s_L + w * WS + 12 * f);
      double qv = 0.0;
      if (jt != GTO_JOINT_FIXED) qv = j >= 0 ? s_qt[(p0 + w) * 8 + j] : s_qp[(p0 + w) * sp.pb_npar + kp];
      float sn = 0.f, cs = 1.f;
      if (jt == GTO_JOINT_REVOLUTE) pb_sincosf(qv, sn, cs);  // Rodrigues (optas/spatialmath.py:90-100); fixed: the identity
      const float c1 = 1.f - cs, u0 = ux.x, u1 = ux.y, u2 = ux.z;
      const float R00 = fmaf(c1 * u0, u0, cs), R01 = fmaf(c1 * u0, u1, -sn * u2), R02 = fmaf(c1 * u0, u2, sn * u1);
      const float R10 = fmaf(c1 * u1, u0, sn * u2), R11 = fmaf(c1 * u1, u1, cs), R12 = fmaf(c1 * u1, u2, -sn * u0);
      const float R20 = fmaf(c1 * u2, u0, -sn * u1), R21 = fmaf(c1 * u2, u1, sn * u0), R22 = fmaf(c1 * u2, u2, cs);
      const float tq = jt == GTO_JOINT_PRISMATIC ? (float)qv : 0.f, t0 = tq * u0, t1 = tq * u1, t2 = tq * u2;
      auto out_row = [&](const gto_f4& o) {
        gto_f4 v;
        v.x = fmaf(o.z, R20, fmaf(o.y, R10, o.x * R00));
        v.y = fmaf(o.z, R21, fmaf(o.y, R11, o.x * R01));
        v.z = fmaf(o.z, R22, fmaf(o.y, R12, o.x * R02));
        v.w = fmaf(o.z, t2, fmaf(o.y, t1, fmaf(o.x, t0, o.w)));
        return v;
      };
      Lo[0] = out_row(o0), Lo[1] = out_row(o1), Lo[2] = out_row(o2);
    }
    __syncthreads();
    if (dbg_t && p0 == 0) bp.dbg[35] = clock64();
    // ---- B: thread (w, row) carries row `row` of the global frame transforms of waypoint p0 + w along the tree (control
    // word of the next frame requested one frame ahead); the global transform of a frame with a moving link overwrites the local
    // transform of its frame (the rows of a waypoint are lanes of one wave: their reads of L_f precede their writes)
    // (Two instances of the walk: robots with at most ONE parked frame -- a Panda's hand, whose second finger is not its
    // successor in the frame order -- select among one register slot instead of GTO_PB_PARK, a third of the instructions of
    // a frame; frames are taken two at a time so that the operands fetched ahead need no copy.)
    auto walk = [&](auto pk_c) {
      constexpr int PK = decltype(pk_c)::value;
      for (int wb = 0; wb < pwc; wb += NT / 4) {
        const int w = wb + (tid >> 2);
        const bool on = w < pwc && row < 3;
        float* __restrict__ Lw = s_L + (on ? w : 0) * WS;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
        float pk[PK][4];
#pragma unroll
        for (int k = 0; k < PK; ++k) pk[k][0] = pk[k][1] = pk[k][2] = pk[k][3] = 0.f;
        auto step = [&](int f, int ctl, const gto_f4& a, const gto_f4& bq, const gto_f4& cq) {
          const int src = GTO_PB_SRC(ctl), pto = GTO_PB_PARKTO(ctl);
          float r0, r1, r2, r3;  // the parent's row
          if (src == 0) {
            r0 = row == 0 ? 1.f : 0.f, r1 = row == 1 ? 1.f : 0.f, r2 = row == 2 ? 1.f : 0.f, r3 = 0.f;
          } else if (src == 1) {
            r0 = g0, r1 = g1, r2 = g2, r3 = g3;
          } else if (PK == 1) {
            r0 = pk[0][0], r1 = pk[0][1], r2 = pk[0][2], r3 = pk[0][3];
          } else {
            r0 = r1 = r2 = r3 = 0.f;
#pragma unroll
            for (int k = 0; k < PK; ++k)
              if (src == 2 + k) r0 = pk[k][0], r1 = pk[k][1], r2 = pk[k][2], r3 = pk[k][3];
          }
          g0 = fmaf(r2, cq.x, fmaf(r1, bq.x, r0 * a.x));
          g1 = fmaf(r2, cq.y, fmaf(r1, bq.y, r0 * a.y));
          g2 = fmaf(r2, cq.z, fmaf(r1, bq.z, r0 * a.z));
          g3 = fmaf(r2, cq.w, fmaf(r1, bq.w, fmaf(r0, a.w, r3)));
#pragma unroll
          for (int k = 0; k < PK; ++k)
            if (pto == k + 1) pk[k][0] = g0, pk[k][1] = g1, pk[k][2] = g2, pk[k][3] = g3;
          if (GTO_PB_LINK(ctl)) {  // global transform of a frame that carries a moving link, into the frame's slot (the chunk
            gto_f4 v;              // centres are kept in the frame's coordinates: PbChunk)
            v.x = g0, v.y = g1, v.z = g2, v.w = g3;
            __builtin_amdgcn_wave_barrier();
            if (on) *reinterpret_cast<gto_f4*>(Lw + 12 * f + 4 * row) = v;
          }
        };
        // a frame's operands are requested one frame ahead: the local transform from LDS, the control word by lane number
        const gto_f4* __restrict__ Lf0 = reinterpret_cast<const gto_f4*>(Lw);
        gto_f4 a0 = Lf0[0], b0 = Lf0[1], c0 = Lf0[2], a1, b1, c1;
        int ctl0 = __builtin_amdgcn_readlane(ctlv, 0), ctl1;
        for (int f = 0; f < F; f += 2) {
          {
            const int fn = min(f + 1, F - 1);
            const gto_f4* __restrict__ Ln = reinterpret_cast<const gto_f4*>(Lw + 12 * fn);
            a1 = Ln[0], b1 = Ln[1], c1 = Ln[2];
            ctl1 = __builtin_amdgcn_readlane(ctlv, fn);
          }
          step(f, ctl0, a0, b0, c0);
          if (f + 1 < F) {
            const int fn = min(f + 2, F - 1);
            const gto_f4* __restrict__ Ln = reinterpret_cast<const gto_f4*>(Lw + 12 * fn);
            a0 = Ln[0], b0 = Ln[1], c0 = Ln[2];
            ctl0 = __builtin_amdgcn_readlane(ctlv, fn);
            step(f + 1, ctl1, a1, b1, c1);
          }
        }
      }
    };
    if (rb->n_xst <= 1) walk(std::integral_constant<int, 1>());
    else walk(std::integral_constant<int, GTO_PB_PARK>());
    __syncthreads();
    if (dbg_t && p0 == 0) bp.dbg[36] = clock64();
    // ---- C: sphere tests, one lane per (waypoint of the pass, chunk of a moving link): the test of k_obstacle_gram's broad
    // phase with the widened radius; a few tests per lane at a time so that their distance-field lookups are in flight together
    const int n_test = pwc * Cm;
    const float box = (float)(bx - sc.ox), boy = (float)(by - sc.oy), boz = (float)(bz - sc.oz), rinv_f = (float)sc.rinv;
    constexpr int PBU = GTO_PB_BATCH;
#pragma unroll 1
    for (int i0 = tid; i0 < n_test; i0 += PBU * NT) {
      int off[PBU], pkd[PBU];  // packed: culling radius | group << 8 | field << 16 | live << 17
#pragma unroll
      for (int u = 0; u < PBU; ++u) {
        const int idx = i0 + u * NT;
        const bool live = idx < n_test;
        const int ic = live ? idx : 0;
        const int w = fast_div(ic, Cm, sp.pb_mC), c = ic - w * Cm;
        const gto_f4 ck = s_ck[c];                      // centre, culling radius (voxels)
        const int cfr = s_ckf[c];                       // frame that carries the chunk's link
        const gto_f4* __restrict__ V = reinterpret_cast<const gto_f4*>(s_L + w * WS + 12 * cfr);
        const gto_f4 va = V[0], vb = V[1], vc = V[2];
        const float u0 = (fmaf(va.z, ck.z, fmaf(va.y, ck.y, fmaf(va.x, ck.x, va.w))) + box) * rinv_f;
        const float u1 = (fmaf(vb.z, ck.z, fmaf(vb.y, ck.y, fmaf(vb.x, ck.x, vb.w))) + boy) * rinv_f;
        const float u2 = (fmaf(vc.z, ck.z, fmaf(vc.y, ck.y, fmaf(vc.x, ck.x, vc.w))) + boz) * rinv_f;
        const int k0 = (int)floorf(u0), k1 = (int)floorf(u1), k2 = (int)floorf(u2);
        const int c0 = min(max(k0, 0), sc.nx - 1), c1 = min(max(k1, 0), sc.ny - 1), c2 = min(max(k2, 0), sc.nz - 1);
        off[u] = c2 + sc.nz * (c1 + sc.ny * c0);
        pkd[u] = (int)ck.w | (((p0 + w) / TG) << 8) | ((p0 + w + 2 >= sp.ts ? 1 : 0) << 16) | ((live ? 1 : 0) << 17);
        asm volatile("" : "+v"(off[u]), "+v"(pkd[u]) :: "memory");  // one test's operands at a time in registers: its two integers are final here
      }
      int dd[PBU];
#pragma unroll
      for (int u = 0; u < PBU; ++u)  // (R >= GTO_DIST_CAP: never culled, dd = 0; the target object's own field from the standoff waypoint on)
        dd[u] = ((pkd[u] >> 17) && (pkd[u] & 255) < GTO_DIST_CAP) ? (int)as_global(((pkd[u] >> 16) & 1) ? sc.d_obs : sc.d_all)[off[u]] : 0;
#pragma unroll
      for (int u = 0; u < PBU; ++u)
        if ((pkd[u] >> 17) && dd[u] <= (pkd[u] & 255)) s_flag[(pkd[u] >> 8) & 255] = 1;
    }
    __syncthreads();
    if (dbg_t && p0 == 0) bp.dbg[37] = clock64();
  }
  if (dbg_t) bp.dbg[38] = clock64();
  // settled groups: the static links' constant as the sum of c^2 of their waypoints, block flags cleared
  {
    const double ss_a = bp.ss_fixed[4 * b + 2], ss_o = bp.ss_fixed[4 * b + 3];
    double* __restrict__ wr = bp.wrec + (((size_t)set_n * B + b) * T + 2) * 8;
    for (int i = tid; i < m * 8; i += NT) {  // eight lanes per waypoint: whole records
      const int sI = i >> 3;
      if (!s_flag[sI / TG]) wr[i] = (i & 7) == 0 ? 0.0 + (sI + 2 < sp.ts ? ss_a : ss_o) : 0.0;
    }
  }
  if (tid < 64) {  // the groups that have to be looked at: one list entry each
    const bool hit = tid < nG && s_flag[tid] != 0;
    const bool on = tid < nG && (hit || sp.pb_verify);
    const unsigned long long mask = __ballot(on);
    int base = 0;
    if (lane == 0 && mask) base = atomicAdd(bp.nlive + 8 + (1 - par), __popcll(mask));
    base = __shfl(base, 0, 64);
    if (on) bp.items[(size_t)(1 - par) * ((size_t)bp.cap * sp.kcap * (T - 2) + GTO_ITEM_SLACK) + base + __popcll(mask & ((1ull << lane) - 1ull))] =
                make_int2(GTO_JOB_ENTRY(b, set_n, 0), (pos_next << 8) | tid | (hit ? 0 : 0x80));
  }
  if (dbg_t) bp.dbg[39] = clock64();
}

// Emptiness certificates ahead of the obstacle launch, for the launches with FEW instances in flight (k_lm_step<8, 4>;
// DESIGN.md section 5).  There the obstacle launch is laid out over every waypoint group of every job -- 1000-2000
// workgroups for 60-190 instances, two dispatch waves of them, four fifths of which run entry -> kinematics -> broad phase
// to find nothing -- and running the exact sphere tests in the step kernel (prebroad_tail) costs the step what it saves the
// evaluation (measured in round 5).  What the stragglers of a call do in those rounds is creep: steps of a millimetre.
// So the LOOK leaves behind how much room it found (k_obstacle_gram<.., ROOM>: per (waypoint, link), the metres every point
// of the link may move before a bounding sphere of it could reach a non-zero voxel record), and this tail only subtracts
// what a candidate's step can move a point of the link by, sum_j reach_link[l][j] |dq_j| (RobotDev::reach_link: the
// farthest a point of link l gets from the axis of joint j, over all configurations; 1 for a prismatic joint).  A group
// all of whose (waypoint, link) pairs keep room is SETTLED: its records (the static links' constant, flag 0) and what is
// left of the room go to the candidate's block set and it gets no workgroup; the others go on the item list the next
// obstacle launch is laid out over, and that launch renews their rooms.  Exact: a culled sphere contributes exact zeros,
// and the bound is the triangle inequality (points within rho of the sphere's centre at the look, moved by at most delta:
// within rho + delta of it now, i.e. within R + ceil(delta / res) index steps of its voxel, which is d > R + room away).
// Rooms are relative to the CURRENT iterate's set (`slot`); `room_ok` says whether that set has them all.
//   s_xs   [KC][m][8] projected steps of the candidates (LDS)      s_work  dead LDS (the systems' region): flags, reach, rooms
template <int NT, int KC>
__device__ __forceinline__ void cert_tail(const RobotDev* __restrict__ rb, const BatchPtrs& bp, const SolveParams& sp, int B, int b, int slot, int NS,
                                          int Kg, bool room_ok, int job0, const double* __restrict__ s_xs, double* __restrict__ s_work, int tid) {
  const int lane = tid & 63;
  const int T = sp.T, m = T - 2, L = rb->n_links, n = rb->n_opt, TG = sp.cert_tg, nG = sp.cert_ng, par = sp.parity;
  int* __restrict__ s_fail = reinterpret_cast<int*>(s_work);          // [KC][64] per candidate and group: has to be looked at
  double* __restrict__ s_rch = s_work + KC * 32;                      // [L][8]
  float* __restrict__ s_av = reinterpret_cast<float*>(s_rch + GTO_MAX_LINKS * 8);  // [Kg][m][L] what is left of the room
  for (int i = tid; i < KC * 64; i += NT) s_fail[i] = room_ok ? 0 : 1;
  for (int i = tid; i < L * 8; i += NT) {
    const int l = i >> 3, j = i & 7;
    s_rch[i] = (j < n && ((rb->link_anc[l] >> j) & 1u)) ? rb->reach_link[l][j] : 0.0;
  }
  __syncthreads();
  if (room_ok) {
    const float* __restrict__ room_c = bp.room + (((size_t)slot * B + b) * T + 2) * L;
    for (int idx = tid; idx < m * L; idx += NT) {
      const int sI = idx / L, l = idx - sI * L;
      const double rm = (double)room_c[idx];
#pragma unroll
      for (int j = 0; j < KC; ++j)
        if (j < Kg) {
          const double* __restrict__ xj = s_xs + ((size_t)j * m + sI) * 8;
          double d = 0.0;
#pragma unroll
          for (int i = 0; i < 8; ++i) d = fma(fabs(xj[i]), s_rch[l * 8 + i], d);
          // (rounding of the sum and of the float the result is stored as: a part in a million and a nanometre, off the room)
          const double a = rm > 0.0 ? (rm - d * 1.000001 - 1e-9) * 0.999999 : -1.0;
          const bool ok = a > 0.0;
          s_av[((size_t)j * m + sI) * L + l] = ok ? (float)a : -1.f;
          if (!ok) s_fail[j * 64 + sI / TG] = 1;
        }
    }
  }
  __syncthreads();
  const double ss_a = bp.ss_fixed[4 * b + 2], ss_o = bp.ss_fixed[4 * b + 3];
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (j < Kg) {  // uniform over the workgroup
      const int set_j = (slot + 1 + j) % NS;
      if (room_ok) {
        double* __restrict__ wr = bp.wrec + (((size_t)set_j * B + b) * T + 2) * 8;
        for (int i = tid; i < m * 8; i += NT) {  // eight lanes per waypoint: whole records
          const int sI = i >> 3;
          if (!s_fail[j * 64 + sI / TG]) wr[i] = (i & 7) == 0 ? 0.0 + (sI + 2 < sp.ts ? ss_a : ss_o) : 0.0;
        }
        float* __restrict__ room_n = bp.room + (((size_t)set_j * B + b) * T + 2) * L;
        for (int idx = tid; idx < m * L; idx += NT)
          if (!s_fail[j * 64 + (idx / L) / TG]) room_n[idx] = s_av[(size_t)j * m * L + idx];
      }
      if (tid < 64) {  // the groups that have to be looked at: one list entry each (debug builds, pb_verify: all, the settled ones marked)
        const bool hit = tid < nG && s_fail[j * 64 + tid] != 0;
        const bool on = tid < nG && (hit || sp.pb_verify);
        const unsigned long long mask = __ballot(on), hitm = __ballot(hit);
        (void)hitm;
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(bp.nlive + 8 + (1 - par), __popcll(mask));
        base = __shfl(base, 0, 64);
        if (on) bp.items[(size_t)(1 - par) * ((size_t)bp.cap * sp.kcap * (T - 2) + GTO_ITEM_SLACK) + base + __popcll(mask & ((1ull << lane) - 1ull))] =
                    make_int2(GTO_JOB_ENTRY(b, set_j, j), ((job0 + j) << 8) | tid | (hit ? 0 : 0x80));
#ifndef GTO_DEBUG_LONGEST_WG
        if (bp.dbg && lane == 0) {  // (GTO_DEBUG_TIMING: groups listed / groups in all, by round of the call)
          atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 128 + min(sp.round, 63)), (unsigned long long)__popcll(hitm));
          atomicAdd(reinterpret_cast<unsigned long long*>(bp.dbg + 192 + min(sp.round, 63)), (unsigned long long)nG);
        }
#endif
      }
    }
}
// doubles of dead LDS cert_tail needs for KL candidates of m free waypoints (the step kernel's systems' region holds KL m 64)
__host__ __device__ inline size_t cert_tail_doubles(int KC, int KL, int m, int L) { return (size_t)KC * 32 + GTO_MAX_LINKS * 8 + ((size_t)KL * m * L + 1) / 2; }

// raw != 0: take Q0's optimised rows as they are (evaluation entry points); otherwise build the seed.
template <int NP>
__global__ __launch_bounds__(256) void k_lm_init(const RobotDev* __restrict__ rb, BatchPtrs bp, SolveParams sp, int B, int raw) {
  const int b = blockIdx.x + bp.b0, tid = threadIdx.x, bl = blockIdx.x;  // bl: the instance's number within its lane
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
    st->pred[0] = 0.0;
    st->ncand = 1;
    st->cflags = 0;
    st->nz0 = st->nz1 = 0ull;
    st->slot = 0;
    st->first = 1;
    st->done = 0;
    st->status = GTO_STATUS_MAX_ITER;
    st->evals = 0;
    st->argmin_cur = 0;
    st->room_ok = st->cand_rooms = 0;
  }
  if (bp.live && tid == 0) {  // the lane's first w0 instances are in flight from round 0 on; the rest wait for one of them to finish
    if (bl < bp.w0) bp.live[bl] = b, bp.jobs[bl] = GTO_JOB_ENTRY(b, 1, 0);  // candidate 0 into set slot + 1
    if (bl == 0) {
      *bp.next = bp.b0 + bp.w0;
      bp.nlive[GTO_NLIVE(0)] = bp.nlive[GTO_NJOBS(0)] = bp.w0;
      bp.nlive[GTO_NLIVE(1)] = bp.nlive[GTO_NJOBS(1)] = 0;
      bp.nlive[8] = bp.nlive[9] = 0;  // round 0 looks at every group of every seed: no item list
    }
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
      else v = v < rb->lower[j] ? rb->lower[j] : (v > rb->upper[j] ? rb->upper[j] : v);  // (a NaN stays a NaN: fmin / fmax would turn it into a limit, and the solve would quietly start somewhere else)
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
      if (bp.live && bl < bp.w0) bp.qfs[(size_t)bl * T * F + idx] = v;  // round 0: job bl
    }
  }
  if (tid < 64) trial_goal_terms_wave<NP>(rb, bp, sp, B, b, tid, 1, 0, st, s_q, s_fr, s_gaff, s_gscr);
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
// every lane (r, c) of the 8x8 lane grid (lane = 8 r + c) <- v of lane (r, J), J a compile-time index: quad broadcast,
// then the other quad of the 8-lane group by row_half_mirror (written into the banks = quads that need it only).
// Two DPP moves per half instead of a trip through the LDS crossbar (ds_bpermute).  The same was tried for the pivot ROW
// (row_ror:8 + v_permlane16_swap + v_permlane32_swap): three dependent levels of 30 cycles each lose against one
// ds_bpermute of 60 (tools/probes/fp64_latency_probe.hip); the row keeps __shfl.
template <int J>
__device__ __forceinline__ double bcast_col8(double v) {
  constexpr int q = J & 3, QP = q | (q << 2) | (q << 4) | (q << 6);
  constexpr int BM = (J >> 2) ? 0x5 : 0xA;  // quads whose own copy is not the source
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), QP, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), QP, 0xf, 0xf, false);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, BM, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, BM, false);
  return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ void gj_pivot8(double& S, int& bad, int r, int c) {
  const double pjj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(S), J * 9),
                                      __builtin_amdgcn_readlane(__double2loint(S), J * 9));
  const double prj = bcast_col8<J>(S);
  const double pjc = __shfl(S, J * 8 + c, 64);
  if (!(pjj > 0.0)) bad = 1;
  const double piv = fast_rcp(pjj);
  const double tcol = -prj * piv;
  const double in_row = (c == J) ? piv : pjc * piv;           // pivot row
  const double off_row = (c == J) ? tcol : fma(tcol, pjc, S);  // other rows
  S = (r == J) ? in_row : off_row;  // selects, not branches
}
// In-place inverse of the SPD 8x8 block held one entry per lane (all 64 lanes active): Gauss-Jordan without pivoting.
// n: optimised joints; rows and columns n .. 7 are padding (rows of the identity nothing couples to): a pivot on one of
// them changes nothing and is not run (same values; a seven-joint arm saves one pivot in eight).
__device__ __forceinline__ int gj_invert8(double& S, int r, int c, int n = 8) {
  int bad = 0;
  gj_pivot8<0>(S, bad, r, c);
  if (n > 1) gj_pivot8<1>(S, bad, r, c);
  if (n > 2) gj_pivot8<2>(S, bad, r, c);
  if (n > 3) gj_pivot8<3>(S, bad, r, c);
  if (n > 4) gj_pivot8<4>(S, bad, r, c);
  if (n > 5) gj_pivot8<5>(S, bad, r, c);
  if (n > 6) gj_pivot8<6>(S, bad, r, c);
  if (n > 7) gj_pivot8<7>(S, bad, r, c);
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

// dynamic LDS layout of k_lm_step (doubles): Z [KL][m][64] | x [KL][m][8] | y [m][8] | e [m][8] | bfull [m][8] |
// Q [8][T] | gaff [16] | red [96] ; then int act [m]   (KL = candidates the launch is laid out for)
// (red: 56 doubles hold everything one candidate's launch keeps there, 96 four candidates'; the active-set masks are one int
// per waypoint.  At the reference's T = 50 the one-candidate launch then needs 40 832 B + 16 static: FOUR workgroups per
// CU instead of three -- until round 5 it asked for 42 560 B (96 doubles and eight ints per waypoint, most of them
// unused), and bench.py --T 48 against --T 50 showed the step between the two: 495 k against 480 k trajectories/s.)
__host__ __device__ inline size_t lm_red_doubles(int KL) { return KL == 1 ? 56 : 96; }
__host__ __device__ inline size_t lm_lds_bytes(int T, int KL) {
  const size_t m = (size_t)T - 2;
  const size_t dbl = (size_t)KL * (m * 64 + m * 8) + 3 * m * 8 + 8 * (size_t)T + 16 + lm_red_doubles(KL);
  return dbl * sizeof(double) + ((m + 1) & ~(size_t)1) * sizeof(int);
}
// One workgroup of NW wavefronts per live instance.  Lane (r,c) = (lane>>3, lane&7) of a wave owns entry (r,c) of the 8x8
// blocks; the data-parallel phases (assembly, projected step, predicted decrease) are spread over the waves by waypoint,
// the serial block recursion of candidate j runs on waves 2j (downwards) and 2j+1 (upwards).
// NW = 4, KC = 1: one candidate per round (launches with many instances in flight).  NW = 8, KC = 4: launches with few
// instances in flight: up to four candidate trial points (dampings lambda, lambda nu, lambda nu 2nu, ...) are generated at
// once, each by its own pair of waves, and the next launch walks their evaluations in the order the sequential algorithm
// would have produced them.  The results do not depend on NW or on how many candidates a launch generates: the only
// cross-wave sum, the predicted decrease, is formed per waypoint class (s mod 8) in ascending order and the eight classes
// are added in a fixed order.
// (launch bounds: the four-wave variant is pinned to four waves per SIMD = 128 VGPRs.  It sits exactly there, and one register
// more -- from a change anywhere in the headers -- takes the evaluation kernel's workgroup off the CU the three step
// workgroups share with it: 3 x 128 + 96 <= 512 registers per lane; measured at 143: default run 475 -> 466 k.)
template <int NW, int KC>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 4 : 1) void k_lm_step(const RobotDev* __restrict__ rb, const struct PbChunk* __restrict__ pbc, BatchPtrs bp, SolveParams sp, int B) {
  static_assert(NW >= 2 * KC, "the twisted factorisation of a candidate takes two wavefronts");
  static_assert(KC <= GTO_KSPEC, "candidate copies of the workspace");
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (GTO_STEP_PRIO != 0) __builtin_amdgcn_s_setprio(GTO_STEP_PRIO);
  if (blockIdx.x == 0 && threadIdx.x == 0 && bp.progress) {  // lagged by design: what had finished when this launch started
    const int nd = __hip_atomic_load(bp.n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(bp.progress, bp.progress_tag | (unsigned long long)(unsigned)nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(bp.progress + 1, bp.progress_tag | (unsigned long long)(unsigned)sp.round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // how many (job, group) pairs this round's obstacle launch had to look at, of how many jobs (both saturating): the host
    // stops asking for the broad phase ahead of the launch in a call where it settles next to nothing
    if (bp.items) __hip_atomic_store(bp.progress + 2, bp.progress_tag | ((unsigned long long)min((unsigned)bp.nlive[GTO_NJOBS(sp.parity)], 0xfffu) << 20) | (unsigned long long)min((unsigned)bp.nlive[8 + sp.parity], 0xfffffu),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // solve loop only: one workgroup per position of this round's live list
  const int par = sp.parity, pos_id = blockIdx.x, NS = sp.kcap + 1;
  const int n_live = bp.nlive[GTO_NLIVE(par)], le_ = bp.live[par * bp.cap + min(pos_id, bp.cap - 1)];  // one round trip for both
  const bool stat = KC == 1 && sp.static_pos != 0;
  if (stat && pos_id == 0 && threadIdx.x == 0) bp.nlive[GTO_NLIVE(1 - par)] = bp.nlive[GTO_NJOBS(1 - par)] = n_live;  // the next lists span the same positions
  auto void_position = [&]() {  // static positions: a position without an instance is void in the next lists, too
    if (stat && threadIdx.x == 0 && pos_id < n_live) bp.live[(1 - par) * bp.cap + pos_id] = -1, bp.jobs[(1 - par) * bp.cap * sp.kcap + pos_id] = -1;
  };
  if (pos_id >= n_live || le_ < 0) {
    void_position();
    return;
  }
  const int b = le_;
  __shared__ int s_nid, s_pos;
  InstState* st = bp.state + b;
  // the whole state in one round trip (the walk over the candidates below is then arithmetic only)
  const int st_done = st->done, first = st->first, slot0 = st->slot, st_ncand = st->ncand, cflags = st->cflags;
  const int st_room_ok = KC > 1 ? st->room_ok : 0, st_cand_rooms = KC > 1 ? st->cand_rooms : 0;  // (emptiness certificates: few-instance launches only)
  double st_pred[KC], st_fg[KC], st_fv[KC];
  int st_am[KC];
#pragma unroll
  for (int j = 0; j < KC; ++j) st_pred[j] = st->pred[j], st_fg[j] = st->fgoal_try[j], st_fv[j] = st->fvel_try[j], st_am[j] = st->argmin_try[j];
  double f = st->f, lambda = st->lambda, nu = st->nu;
  const unsigned long long st_nz0 = st->nz0, st_nz1 = st->nz1;
  int status = st->status, k = st->evals, argmin_cur = st->argmin_cur;
  const double fixed = bp.ss_fixed[4 * b] + bp.ss_fixed[4 * b + 1];
  // (the instance's scene number, for the tail: asked for here, with the state, it is one trip to memory less in front of
  // the tail's tables, whose culling radii wait for the scene's voxel size)
  int scene_b = (KC == 1 && sp.pb_next) ? bp.scene_id[b] : 0;
  asm volatile("" : "+v"(scene_b));
  if (st_done) {  // an instance that finished after it had been listed (late exits of the step kernel)
    void_position();
    return;
  }
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int T = sp.T, n = rb->n_opt, m = T - 2;
  const int KL = min(KC, max(sp.k_acc, sp.k_rej));  // candidates this launch's LDS is laid out for
  double* s_Z = smem;                        // [KL][m][64]
  double* s_x = s_Z + (size_t)KL * m * 64;   // [KL][m][8]
  double* s_y = s_x + (size_t)KL * m * 8;
  double* s_e = s_y + m * 8;
  double* s_b = s_e + m * 8;
  double* s_Q = s_b + m * 8;
  double* s_gaff = s_Q + 8 * T;
  // [96] cross-wave scratch: [2j], [2j+1] failure flags of candidate j's two sweeps, [8] first dense block (int, early),
  // [16 + 8j + w] step maximum of candidate j seen by wave w, [48 + 8j + c] predicted decrease of candidate j by class
  double* s_red = s_gaff + 16;
  int* s_actm = (int*)(s_red + lm_red_doubles(KL));  // [m] frozen-joint bit masks
  int* s_first_dense = (int*)(s_red + 8);    // first waypoint block with an off-diagonal entry

  const int r = lane >> 3, c = lane & 7;
  const long long t_dbg0 = bp.dbg ? clock64() : 0;
  // P4 writes joint (tid & 7) of some waypoints: its frame, looked up long before it is needed
  const int nF = rb->n_frames, my_frame = (tid & 7) < n ? rb->opt_frame[tid & 7] : 0;
  // ... and its limits: every (waypoint, joint) item of this thread has joint index tid & 7
  const double my_lo = (tid & 7) < n ? rb->lower[tid & 7] : 0.0, my_hi = (tid & 7) < n ? rb->upper[tid & 7] : 0.0;

  // ---- P0: objective of the candidates evaluated last round (every wave computes them: cheaper than a broadcast); the
  // loads of all candidate sets go out together, whether the candidate exists or not
  const int ncand = first ? 1 : min(st_ncand, KC);
  double f_try[KC];
  unsigned long long nzm0[KC + 1], nzm1[KC + 1];  // non-zero blocks of set slot0 + j (j = 0: the current iterate's)
  {
    static_assert(GTO_MAX_T - 2 <= 128, "two waypoints per lane");
    double fo[KC];
    bool zc0[KC], zc1[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const double2* __restrict__ wr = reinterpret_cast<const double2*>(bp.wrec + ((size_t)((slot0 + 1 + j) % NS) * B + b) * T * 8);
      const double2 r0_ = 2 + lane < T ? wr[(size_t)(2 + lane) * 4] : make_double2(0.0, 0.0);
      const double2 r1_ = 66 + lane < T ? wr[(size_t)(66 + lane) * 4] : make_double2(0.0, 0.0);
      fo[j] = r0_.x + r1_.x;
      zc0[j] = r0_.y != 0.0, zc1[j] = r1_.y != 0.0;
    }
    // which blocks of the candidate sets are non-zero: bit s of nzm0 = waypoint s + 2, of nzm1 = waypoint s + 66; the flags
    // ride in the records of the sums (no further load), the current set's came with the instance's state
    nzm0[0] = st_nz0, nzm1[0] = st_nz1;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      nzm0[j + 1] = __ballot(zc0[j]);
      nzm1[j + 1] = __ballot(zc1[j]);
    }
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      double fs = wave_sum(fo[j]);
      fs += fixed;
      f_try[j] = st_fg[j] + sp.w_obstacle * fs + st_fv[j];
    }
  }

  // ---- P1: walk the candidates in the order the sequential algorithm would have met them (oracle/gto_oracle.c
  // solve_instance: evaluate, accept or reject, stopping tests, next trial point); all threads take the same branches
  int slot = slot0, done = 0, acc = first ? 0 : -1;
  {
    bool go = !first;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      if (go && j < ncand) {
        if (j > 0) {  // what generating candidate j after the rejection of j - 1 would have found
          if ((cflags >> j) & 1) {
            status = GTO_STATUS_NUMERICAL;
            done = 1;
            go = false;
          } else if ((cflags >> (8 + j)) & 1) {
            status = GTO_STATUS_CONVERGED;
            done = 1;
            go = false;
          } else {
            ++k;
          }
        }
        if (go) {
          const double pj = st_pred[j], ft = f_try[j];
          if (ft < f && pj > 0.0) {
            const double df = f - ft, rho = df / pj;
            const double sg = 2.0 * rho - 1.0;
            double fac = 1.0 - sg * sg * sg;
            fac = fmax(fac, 1.0 / 3.0);
            lambda = fmax(lambda * fac, 1e-12);
            nu = 2.0;
            if (df <= sp.tol_rel_f * (1.0 + ft)) {
              status = GTO_STATUS_CONVERGED;
              done = 1;
            }
            acc = j;
            go = false;
          } else {
            lambda *= nu;
            nu *= 2.0;
            if (lambda > 1e15) {
              status = GTO_STATUS_CONVERGED;
              done = 1;
              go = false;
            } else if (k >= sp.max_iter) {
              status = GTO_STATUS_MAX_ITER;
              done = 1;
              go = false;
            }
          }
        }
      }
    }
  }
  const bool accept = acc >= 0;
  double* __restrict__ Qc = bp.Qcur + (size_t)b * n * T;
  if (accept) {
#pragma unroll
    for (int j = 0; j < KC; ++j)
      if (j == acc) {
        f = f_try[j];
        argmin_cur = st_am[j];
      }
    slot = (slot0 + 1 + acc) % NS;
  }
  unsigned long long nz0 = nzm0[0], nz1 = nzm1[0];  // non-zero blocks of the set that is current from here on
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (accept && j == acc) nz0 = nzm0[j + 1], nz1 = nzm1[j + 1];
  auto blk_nz = [&](int s_) -> bool { return s_ < 64 ? (nz0 >> s_) & 1ull : (nz1 >> (s_ - 64)) & 1ull; };
  // the iterate that is current from here on: loaded first (loads return in order), copied below
  constexpr int NQ = (8 * GTO_MAX_T + NT - 1) / NT;
  double qv[NQ];
  {
    const double* __restrict__ src = accept ? bp.Qtry + ((size_t)acc * B + b) * n * T : Qc;
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
    av[kk] = (inb && s < m && blk_nz(s)) ? oblk[(size_t)(s + 2) * BLK_STRIDE + BLK_JTJ + lane] : 0.0;
  }
  double jv[NU];  // obstacle J^T r of this thread's (waypoint, joint) items
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int idx = tid + NT * u, i = idx & 7;
    jv[u] = (idx < m * 8 && i < n && blk_nz(idx >> 3)) ? oblk[(size_t)((idx >> 3) + 2) * BLK_STRIDE + BLK_JTR + i] : 0.0;
  }
  const double gA0 = inb ? gblk[BLK_JTJ + lane] : 0.0;
  const double gA1 = (inb && sp.use_standoff) ? gblk[BLK_STRIDE + BLK_JTJ + lane] : 0.0;
  double gjv = 0.0;  // goal J^T r of both goal waypoints (threads 0..15), parked in LDS in P2
  if (tid < 16) {
    const int w = tid >> 3, i = tid & 7;
    if (i < n && (w == 0 || sp.use_standoff)) gjv = gblk[w * BLK_STRIDE + BLK_JTR + i];
  }
  // joint values of the frames no optimised joint drives (and of the two pinned waypoints): they follow the instance to its
  // position in the next round's list; k_lm_init left them in bp.qf
  constexpr int NFV = 4;
  double fv_[NFV];
  unsigned fv_keep = 0;  // bit u: entry tid + NT u is one of those (P4 writes the others: optimised joints, free waypoints)
  {
    const double* __restrict__ qf0 = bp.qf + (size_t)b * T * nF;
#pragma unroll
    for (int u = 0; u < NFV; ++u) {
      const int idx = tid + NT * u;
      fv_[u] = idx < T * nF ? qf0[idx] : 0.0;
      if constexpr (KC == 1) {
        // (the four-wave variant: the frame of an entry by the division's magic number and a bit of a mask out of the robot
        // table -- a remainder by a run-time divisor is forty instructions and the look-up it fed a dependent trip in front
        // of the loads below; the eight-wave variant's schedule is the worse for the same change: measured, A.0)
        const int fr_ = idx - fast_div(idx, nF, sp.pb_mF) * nF;
        if (idx < T * nF && (idx < 2 * nF || ((rb->frame_free_mask >> fr_) & 1u))) fv_keep |= 1u << u;
      } else {
        if (idx < T * nF && (idx < 2 * nF || rb->opt_of_frame[idx % nF] < 0)) fv_keep |= 1u << u;
      }
    }
  }
  // current iterate into LDS (rows >= n are padding); on accept it is the accepted candidate
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
  // a seed whose objective is not a finite number (oracle/gto_oracle.c solve_instance): the iterate goes back as it is
  if (first && !(fabs(f) < INFINITY)) {
    status = GTO_STATUS_NUMERICAL;
    done = 1;
  }
  // thread 0: this instance's position in the next round's lists, once it has one (static positions: its own, from the start)
  int pos_t0 = stat ? pos_id : -1, job_t0 = stat ? pos_id : -1, njob_t0 = stat ? 1 : 0;
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
      st->evals = k;                              \
      st->argmin_cur = argmin_cur;                \
      atomicAdd(bp.n_done, 1);                    \
      /* the next instance of the call that has not started takes over (the places this one had already taken in \
         the next lists, if any; places left without an instance or a candidate are marked void) */ \
      const int nid_ = atomicAdd(bp.next, 1);     \
      s_nid = nid_ < bp.n_total ? nid_ : -1;      \
      int p_ = pos_t0, q_ = job_t0;               \
      if (s_nid >= 0 && p_ < 0) {                 \
        const unsigned long long pq_ = atomicAdd(reinterpret_cast<unsigned long long*>(bp.nlive + GTO_NLIVE(1 - par)), 1ull | (1ull << 32)); \
        p_ = (int)(unsigned)pq_, q_ = (int)(pq_ >> 32); \
      }                                           \
      if (p_ >= 0) {                              \
        bp.live[(1 - par) * bp.cap + p_] = s_nid; \
        bp.jobs[(1 - par) * bp.cap * sp.kcap + q_] = s_nid < 0 ? -1 : GTO_JOB_ENTRY(s_nid, 1, 0); /* candidate 0 of the newcomer (k_lm_init: slot 0), or void */ \
        for (int j_ = 1; j_ < njob_t0; ++j_) bp.jobs[(1 - par) * bp.cap * sp.kcap + q_ + j_] = -1; \
      }                                           \
      s_pos = q_;                                 \
    }                                             \
    __syncthreads();                              \
    { /* ... with the joint values of its seed (k_lm_init left them in qf) */ \
      const int nid = s_nid;                      \
      if (nid >= 0) {                             \
        double* __restrict__ dst_ = bp.qfs + ((size_t)(1 - par) * bp.cap * sp.kcap + s_pos) * sp.T * rb->n_frames; \
        for (int i_ = tid; i_ < sp.T * rb->n_frames; i_ += NT) dst_[i_] = bp.qf[(size_t)nid * sp.T * rb->n_frames + i_]; \
        if (KC > 1 && sp.cert_next && tid == 0) bp.state[nid].cand_rooms = 1; /* (its seed's look will leave the rooms of every group) */ \
        if ((sp.pb_next || (KC > 1 && sp.cert_next)) && tid < 64) { /* nothing is known about a seed: every group of its job goes on the item list */ \
          const bool on_ = tid < (sp.pb_next ? sp.pb_ng : sp.cert_ng); \
          const unsigned long long mk_ = __ballot(on_); \
          int base_ = 0;                          \
          if (lane == 0) base_ = atomicAdd(bp.nlive + 8 + (1 - par), __popcll(mk_)); \
          base_ = __shfl(base_, 0, 64);           \
          if (on_) bp.items[(size_t)(1 - par) * ((size_t)bp.cap * sp.kcap * (sp.T - 2) + GTO_ITEM_SLACK) + base_ + tid] = make_int2(GTO_JOB_ENTRY(nid, 1, 0), (s_pos << 8) | tid); \
        }                                         \
      }                                           \
    }                                             \
    return;                                       \
  } while (0)
  __syncthreads();
  if (done) GTO_FINISH(status);

  // candidates of this step: K_g dampings lam[0] = lambda, lam[j+1] = lam[j] nu_j, nu_{j+1} = 2 nu_j (what the sequential
  // algorithm would use after j rejections in a row); none beyond the stopping tests that would end the solve first
  const int streak = acc == 0 ? min(((cflags >> 16) & 0xff) + 1, 255) : 0;
  int Kg = accept ? ((sp.spec_streak > 0 && streak >= sp.spec_streak) ? 1 : sp.k_acc) : sp.k_rej;
  Kg = max(1, min(min(Kg, KL), min(sp.kcap, sp.max_iter - k)));
  double lam[KC];
  lam[0] = lambda;
  {
    double l_ = lambda, v_ = nu;
#pragma unroll
    for (int j = 1; j < KC; ++j) {
      l_ *= v_;
      v_ *= 2.0;
      lam[j] = l_;
      if (j < Kg && l_ > 1e15) Kg = j;
    }
  }
  if (tid == 0) {  // this instance goes on: its place in the next round's list, its candidates' places in the job list
    // (one add for both lists: every workgroup of the launch adds to the same counters, and they are served one by one)
    if (!stat) {
      const unsigned long long pq = atomicAdd(reinterpret_cast<unsigned long long*>(bp.nlive + GTO_NLIVE(1 - par)), 1ull | ((unsigned long long)Kg << 32));
      pos_t0 = (int)(unsigned)pq, job_t0 = (int)(pq >> 32);
      njob_t0 = Kg;
    }
    bp.live[(1 - par) * bp.cap + pos_t0] = b;
    for (int j = 0; j < Kg; ++j) bp.jobs[(1 - par) * bp.cap * sp.kcap + job_t0 + j] = GTO_JOB_ENTRY(b, (slot + 1 + j) % NS, j);
    s_pos = job_t0;
  }

  if (bp.dbg && b == 0 && tid == 0) {
    bp.dbg[0] = t_dbg0;  // (with the P2 stamp, not at the end: a launch that leaves through GTO_FINISH writes neither)
    bp.dbg[1] = clock64();
  }
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
  const int pos_next = s_pos;
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[29] = clock64();
  // undamped entries stay in registers (av[kk], read again by P5: same wave, same waypoints), the damped /
  // frozen systems of the candidates go to s_Z; remember which blocks are purely diagonal.  Everything that depends on
  // the lane only is hoisted; the two waypoints that carry goal terms are patched by the wave that owns them.
  const bool diagl = r == c;
  const double dadd = (inb && diagl) ? 2.0 * alpha : 0.0;  // velocity term of an interior waypoint
  const double idv = diagl ? 1.0 : 0.0;
  const int lane_bits = (1 << r) | (1 << c);
  constexpr unsigned long long kOffDiag = ~0x8040201008040201ull;  // lanes (r,c) with r != c
  {
    int first_d = m;  // first dense block seen by this wave
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
        av[kk] = a;
#pragma unroll
        for (int j = 0; j < KC; ++j)
          if (j < Kg) {
            const double dmul = diagl ? 1.0 + lam[j] : 1.0;
            const double v = frozen ? idv : a * dmul;
            s_Z[((size_t)j * m + s) * 64 + lane] = v;
            if (j == 0 && (__ballot(v != 0.0) & kOffDiag) && s < first_d) first_d = s;
          }
      }
    }
    if (lane == 0 && first_d < m) atomicMin(s_first_dense, first_d);
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
  // factorisation): the even wave of a candidate eliminates downwards from waypoint 0, the odd wave upwards from the last
  // waypoint, they meet at block `mid`, and the two back-substitutions run outwards from there, again in parallel.
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
  auto gj_invert = [&](double& S) -> int { return gj_invert8(S, r, c, n); };
  auto matvec = [&](double Z, double zc) -> double { return matvec8(Z, zc); };
  // candidate and sweep of this wave.  Eight waves sit on four SIMDs (wave w on SIMD w & 3), so from the third candidate
  // on two sweeps share a SIMD: the upward sweep is all dense blocks (bound by instruction issue), the downward one
  // starts with the diagonal stretch (bound by the latency of its chain).  One of each per SIMD, the upward one first in
  // line at the issue port: the barrier behind the sweeps of four candidates after 17.0 K cycles (like on like: 19.0 K)
  const int cj = wave >> 1, role = (wave & 1) ^ ((wave >> 2) & 1);
  const bool solver = cj < Kg && cj < KC;          // wave-uniform
  double* __restrict__ cZ = s_Z + (size_t)(solver ? cj : 0) * m * 64;
  double* __restrict__ cx = s_x + (size_t)(solver ? cj : 0) * m * 8;
  if (solver && role == 0) {
    int fail = 0;
    double zp = 0.0, yp = 0.0;
    const int nd = s_dense < mid ? s_dense : mid;  // diagonal stretch of the downward sweep
    if (r == c) {
      // Every instruction of this wave issues in order, so a step costs the sum of its instructions, not the depth of
      // its chain: the two recurrences run as two tight loops (pivots, then right-hand side), four steps at a time with
      // the operands of the four fetched first and nothing conditional in between.  Same operations in the same order
      // per step as the plain loop that finishes the stretch.
      int s = 0;
      double smin = 1.0;
      for (; s + GTO_DIAG_BATCH <= nd; s += GTO_DIAG_BATCH) {
        double e2[GTO_DIAG_BATCH], d[GTO_DIAG_BATCH];
#pragma unroll
        for (int u = 0; u < GTO_DIAG_BATCH; ++u) {
          const double ep = s_e[(s + u > 0 ? s + u - 1 : 0) * 8 + r];
          e2[u] = (s + u > 0) ? ep * ep : 0.0;
          d[u] = cZ[(size_t)(s + u) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < GTO_DIAG_BATCH; ++u) {
          const double S = d[u] - e2[u] * zp;
          smin = fmin(smin, S);  // NaN-safe below: !(S > 0) for any S shows as !(smin > 0) or as a NaN pivot caught at the end
          zp = fast_rcp(S);
          cZ[(size_t)(s + u) * 64 + lane] = zp;
        }
      }
      const int sb = s;  // steps [0, sb) have their pivots
      if (!(smin > 0.0) || zp != zp) fail = 1;
      for (s = 0; s < sb; s += GTO_DIAG_BATCH) {
        double ep[GTO_DIAG_BATCH], z[GTO_DIAG_BATCH], rh[GTO_DIAG_BATCH];
#pragma unroll
        for (int u = 0; u < GTO_DIAG_BATCH; ++u) {
          ep[u] = s_e[(s + u > 0 ? s + u - 1 : 0) * 8 + r];
          z[u] = cZ[(size_t)(s + u) * 64 + lane];
          rh[u] = s_y[(s + u) * 8 + r];
        }
#pragma unroll
        for (int u = 0; u < GTO_DIAG_BATCH; ++u) {
          yp = z[u] * (rh[u] - ((s + u > 0) ? ep[u] : 0.0) * yp);
          cx[(s + u) * 8 + r] = yp;
        }
      }
      for (s = sb; s < nd; ++s) {
        const double ep = (s > 0) ? s_e[(s - 1) * 8 + r] : 0.0;
        const double S = cZ[(size_t)s * 64 + lane] - ep * ep * zp;
        if (!(S > 0.0)) fail = 1;
        const double Zr = fast_rcp(S);
        const double y = Zr * (s_y[s * 8 + r] - ep * yp);
        cZ[(size_t)s * 64 + lane] = Zr;
        cx[s * 8 + r] = y;
        zp = Zr;
        yp = y;
      }
    }
    wave_sync();
    if (bp.dbg && b == 0 && tid == 0) bp.dbg[3] = clock64();
    double Zprev = (r == c) ? zp : 0.0;
    double yprev_c = (nd > 0) ? cx[(nd - 1) * 8 + c] : 0.0;  // y_{s-1}[c] for lane (r,c)
    for (int s = nd; s < mid; ++s) {
      double S = cZ[(size_t)s * 64 + lane];
      double zc = s_y[s * 8 + c];
      if (s > 0) {
        const double er = s_e[(s - 1) * 8 + r], ec = s_e[(s - 1) * 8 + c];
        S -= er * ec * Zprev;
        zc -= ec * yprev_c;
      }
      fail |= gj_invert(S);
      cZ[(size_t)s * 64 + lane] = S;  // Z_s
      Zprev = S;
      const double pr = matvec(S, zc);
      if (c == 0) cx[s * 8 + r] = pr;
      yprev_c = __shfl(pr, c << 3, 64);  // transpose: lane (r,c) picks y_s[c] from row c
    }
    if (lane == 0) s_red[2 * cj] = __any(fail) ? 1.0 : 0.0;
  } else if (solver) {
    if constexpr (NW == 8) __builtin_amdgcn_s_setprio(GTO_STEP_PRIO > 2 ? GTO_STEP_PRIO : 2);
    int fail = 0;
    double Zprev = 0.0, yprev_c = 0.0;
    for (int s = m - 1; s > mid; --s) {
      double S = cZ[(size_t)s * 64 + lane];
      double zc = s_y[s * 8 + c];
      if (s < m - 1) {
        const double er = s_e[s * 8 + r], ec = s_e[s * 8 + c];
        S -= er * ec * Zprev;
        zc -= ec * yprev_c;
      }
      fail |= gj_invert(S);
      cZ[(size_t)s * 64 + lane] = S;
      Zprev = S;
      const double pr = matvec(S, zc);
      if (c == 0) cx[s * 8 + r] = pr;
      yprev_c = __shfl(pr, c << 3, 64);
    }
    if (lane == 0) s_red[2 * cj + 1] = __any(fail) ? 1.0 : 0.0;
    if constexpr (NW == 8) __builtin_amdgcn_s_setprio(GTO_STEP_PRIO > 2 ? GTO_STEP_PRIO : (GTO_STEP_PRIO ? GTO_STEP_PRIO : 0));
    if (bp.dbg && b == 0 && tid == 64) bp.dbg[17] = clock64();
  }
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[41] = clock64();
  if (bp.dbg && b == 0 && lane == 0 && wave >= 2 && wave < 7) bp.dbg[41 + wave] = solver ? clock64() : 0;
  __syncthreads();
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[4] = clock64();
  if (solver && role == 0) {  // the meeting block
    int fail = (s_red[2 * cj] != 0.0) || (mid < m - 1 && s_red[2 * cj + 1] != 0.0);
    double S = cZ[(size_t)mid * 64 + lane];
    double zc = s_y[mid * 8 + c];
    if (mid > 0) {
      const double er = s_e[(mid - 1) * 8 + r], ec = s_e[(mid - 1) * 8 + c];
      S -= er * ec * cZ[(size_t)(mid - 1) * 64 + lane];
      zc -= ec * cx[(mid - 1) * 8 + c];
    }
    if (mid < m - 1) {
      const double er = s_e[mid * 8 + r], ec = s_e[mid * 8 + c];
      S -= er * ec * cZ[(size_t)(mid + 1) * 64 + lane];
      zc -= ec * cx[(mid + 1) * 8 + c];
    }
    fail |= gj_invert(S);
    const double pr = matvec(S, zc);
    if (c == 0) cx[mid * 8 + r] = pr;  // x_mid
    if (lane == 0) s_red[2 * cj] = __any(fail) ? 1.0 : 0.0;
  }
  __syncthreads();
  if (s_red[0] != 0.0) GTO_FINISH(GTO_STATUS_NUMERICAL);
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[42] = clock64();
  int cflags_new = streak << 16;
#pragma unroll
  for (int j = 1; j < KC; ++j)
    if (j < Kg && s_red[2 * j] != 0.0) cflags_new |= 1 << j;
  if (solver && role == 0) {  // outwards to waypoint 0
    const int nd = s_dense < mid ? s_dense : mid;
    double xr = cx[mid * 8 + r], xc = cx[mid * 8 + c];
    for (int s = mid - 1; s >= nd; --s) {  // dense blocks
      const double pr = matvec(cZ[(size_t)s * 64 + lane], s_e[s * 8 + c] * xc);
      xr = cx[s * 8 + r] - pr;
      if (c == 0) cx[s * 8 + r] = xr;
      xc = __shfl(xr, c << 3, 64);
    }
    if (bp.dbg && b == 0 && tid == 0) bp.dbg[18] = clock64();
    if (r == c) {  // diagonal stretch, four steps at a time as in the downward sweep
      int s = nd - 1;
      for (; s >= GTO_DIAG_BATCH - 1; s -= GTO_DIAG_BATCH) {
        double yv[GTO_DIAG_BATCH], zv[GTO_DIAG_BATCH], ev[GTO_DIAG_BATCH];
#pragma unroll
        for (int u = 0; u < GTO_DIAG_BATCH; ++u) {
          yv[u] = cx[(s - u) * 8 + r];
          zv[u] = cZ[(size_t)(s - u) * 64 + lane];
          ev[u] = s_e[(s - u) * 8 + r];
        }
#pragma unroll
        for (int u = 0; u < GTO_DIAG_BATCH; ++u) {
          xr = yv[u] - zv[u] * (ev[u] * xr);
          cx[(s - u) * 8 + r] = xr;
        }
      }
      for (; s >= 0; --s) {
        xr = cx[s * 8 + r] - cZ[(size_t)s * 64 + lane] * (s_e[s * 8 + r] * xr);
        cx[s * 8 + r] = xr;
      }
    }
    if (bp.dbg && b == 0 && tid == 0) bp.dbg[19] = clock64();
  } else if (solver) {  // outwards to the last waypoint
    double xc = cx[mid * 8 + c];
    for (int s = mid + 1; s < m; ++s) {
      const double pr = matvec(cZ[(size_t)s * 64 + lane], s_e[(s - 1) * 8 + c] * xc);
      const double xr = cx[s * 8 + r] - pr;
      if (c == 0) cx[s * 8 + r] = xr;
      xc = __shfl(xr, c << 3, 64);
    }
    if (bp.dbg && b == 0 && tid == 64) bp.dbg[26] = clock64();
  }
  __syncthreads();

  if (bp.dbg && b == 0 && tid == 0) bp.dbg[5] = clock64();
  // ---- P4: projected trial points; s_x becomes the projected step of each candidate.  The joint values go to the
  // instance's position in the next round's list, a full [T][F] table per candidate (the obstacle kernel reads them by frame)
  double* __restrict__ qfn = bp.qfs + ((size_t)(1 - par) * bp.cap * sp.kcap + pos_next) * T * nF;  // pos_next: first job of this instance
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (j < Kg) {
      double* __restrict__ qfj = qfn + (size_t)j * T * nF;
#pragma unroll
      for (int u = 0; u < NFV; ++u)
        if ((fv_keep >> u) & 1u) qfj[tid + NT * u] = fv_[u];
      for (int idx = tid + NT * NFV; idx < T * nF; idx += NT)  // very large robots
        if (idx < 2 * nF || rb->opt_of_frame[idx % nF] < 0) qfj[idx] = bp.qf[(size_t)b * T * nF + idx];
    }
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (j < Kg) {
      double* __restrict__ Qt = bp.Qtry + ((size_t)j * B + b) * n * T;
      double* __restrict__ qfj = qfn + (size_t)j * T * nF;
      double* __restrict__ xj = s_x + (size_t)j * m * 8;
      double maxstep = 0.0;
      for (int idx = tid; idx < m * 8; idx += NT) {
        const int sI = idx >> 3, i = idx & 7, t = sI + 2;
        double sv = 0.0;
        if (i < n) {
          const double q0 = s_Q[i * T + t];
          double v = q0 + xj[idx];
          v = fmin(fmax(v, my_lo), my_hi);
          Qt[(size_t)i * T + t] = v;
          qfj[(size_t)t * nF + my_frame] = v;
          sv = v - q0;
          if constexpr (KC == 1) s_e[idx] = v;  // the couplings are dead after the solve: the trial point, for the broad phase below
        }
        xj[idx] = sv;
        maxstep = fmax(maxstep, fabs(sv));
      }
      if (tid < n * 2) Qt[(size_t)(tid >> 1) * T + (tid & 1)] = s_Q[(tid >> 1) * T + (tid & 1)];
      maxstep = wave_max(maxstep);
      if (lane == 0) s_red[16 + 8 * j + wave] = maxstep;
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KC; ++j)
    if (j < Kg) {
      double maxstep = s_red[16 + 8 * j];
#pragma unroll
      for (int w = 1; w < NW; ++w) maxstep = fmax(maxstep, s_red[16 + 8 * j + w]);
      if (maxstep < sp.tol_step) {
        if (j == 0) GTO_FINISH(GTO_STATUS_CONVERGED);
        cflags_new |= 1 << (8 + j);
      }
    }

  if (bp.dbg && b == 0 && tid == 0) bp.dbg[6] = clock64();
  // ---- P5: predicted decrease of the undamped model: -(2 b.s + s^T A s), waypoints split over waves.
  // Lane (r,c) adds A[r][c] s_r s_c; the lanes of column 0 also add the terms that are linear in s_r
  // (2 b_r s_r and the coupling -2 alpha s_r s'_r with the next waypoint), folded in as a lane constant.
  {
    static_assert(NW == 4 || NW == 8, "the waypoint classes of the predicted decrease are laid out for 4 or 8 waves");
    const double c0 = (c == 0) ? 2.0 : 0.0;
#pragma unroll
    for (int j = 0; j < KC; ++j)
      if (j < Kg) {
        const double* __restrict__ xj = s_x + (size_t)j * m * 8;
        double part0 = 0.0, part1 = 0.0;  // class wave (and, with four waves, class wave + 4: the odd kk)
#pragma unroll
        for (int kk = 0; kk < KMAX; ++kk) {
          const int s = wave + NW * kk;
          if (s < m) {  // wave-uniform
            const double sr = xj[s * 8 + r], scv = xj[s * 8 + c];
            const double xn = (s < m - 1) ? xj[(s + 1) * 8 + r] : 0.0;
            const double lin = c0 * fma(-alpha, xn, s_b[s * 8 + r]);
            if (NW == 8 || (kk & 1) == 0) part0 = fma(sr, fma(av[kk], scv, lin), part0);
            else part1 = fma(sr, fma(av[kk], scv, lin), part1);
          }
        }
        part0 = wave_sum(part0);
        if (lane == 0) s_red[48 + 8 * j + wave] = part0;
        if (NW == 4) {
          part1 = wave_sum(part1);
          if (lane == 0) s_red[48 + 8 * j + wave + 4] = part1;
        }
      }
  }
  __syncthreads();
  if (tid == 0) {
    st->f = f;
    st->lambda = lambda;
    st->nu = nu;
#pragma unroll
    for (int j = 0; j < KC; ++j)
      if (j < Kg) {
        const double* pr_ = s_red + 48 + 8 * j;
        st->pred[j] = -(((pr_[0] + pr_[1]) + (pr_[2] + pr_[3])) + ((pr_[4] + pr_[5]) + (pr_[6] + pr_[7])));
      }
    st->ncand = Kg;
    st->cflags = cflags_new;
    st->nz0 = nz0, st->nz1 = nz1;
    st->slot = slot;
    st->first = 0;
    st->status = status;
    st->evals = k + 1;
    st->argmin_cur = argmin_cur;
    if constexpr (KC > 1) {
      st->room_ok = accept ? st_cand_rooms : st_room_ok;
      st->cand_rooms = sp.cert_next ? 1 : 0;
    }
  }
  // Emptiness certificates of the next round's jobs of this instance (launches with few instances in flight: cert_tail)
  if constexpr (KC > 1) {
    if (sp.cert_next) {
      __syncthreads();  // P5 is through with the steps' consumers of s_Z (the systems' region is dead since the back substitution)
      cert_tail<NT, KC>(rb, bp, sp, B, b, slot, NS, Kg, (accept ? st_cand_rooms : st_room_ok) != 0, pos_next, s_x, s_Z, tid);
    }
  }
  // Broad phase of the next round's job of this instance (the rounds that fill the GPU; prebroad_tail says what and why):
  // here the trial point is still in LDS, and the obstacle launch then only holds workgroups that have something to gather.
  if constexpr (KC == 1) {
    if (sp.pb_next) {
      __syncthreads();  // P5 is through with the step and the right-hand sides: everything in front of s_gaff is free
      prebroad_tail<NT>(rb, pbc, bp, sp, B, b, scene_b, (slot + 1) % NS, pos_next, s_e, smem, tid);
    }
  }
#undef GTO_FINISH
  if (bp.dbg && b == 0 && tid == 0) bp.dbg[7] = clock64();
  // The goal-set and velocity terms of the new candidates are evaluated by the extra workgroups of the next
  // k_obstacle_gram launch (off the serial path).
  if (bp.dbg && b == 0 && tid == 0) {
    bp.dbg[8] = clock64();
    bp.dbg[9] = s_dense;
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
         2 * 64 + 2 * 8 + 8 + 8 + 8 + GTO_MAX_DOF + 16 + L * 8 + L;
}

#ifndef GTO_IK_PD
#define GTO_IK_PD 2  // surviving chunks of a wave whose points and voxel records are in flight together (k_ik_solve)
#endif
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
  double* s_v = s_red + 16;                    // [L][8]: the four waves' sums of a link, folded in wave order
  int* s_anc = reinterpret_cast<int*>(s_v + L * 8);  // [L]: optimised joints above a link (bit mask)
  const bool collide = scene_id != nullptr;
  {
    const int nt = fk_tab_doubles(F, L, n);
    for (int k = tid; k < nt; k += 256) s_tab[k] = rb->fk_tab[k];
    if (tid < L) s_anc[tid] = (int)rb->link_anc[tid];
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
  // the chunks of this wave (a quarter of the robot's, at most 64): lane i keeps the i-th one's descriptor for the whole solve
  const int nC = rb->n_chunks;
  const int c0 = (int)(((long)nC * wave) / 4), c1 = (int)(((long)nC * (wave + 1)) / 4);
  int ck_link = 0, ck_start = 0, ck_count = 0;
  double ck_cx = 0.0, ck_cy = 0.0, ck_cz = 0.0, ck_r = 0.0;
  if (collide && lane < c1 - c0) {
    const Chunk cc = chunks[c0 + lane];
    ck_link = cc.link, ck_start = cc.start, ck_count = cc.count, ck_cx = cc.cx, ck_cy = cc.cy, ck_cz = cc.cz, ck_r = cc.r;
  }
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
      s_gaff[tid] = Xg[fkx(fsel, 4 * (e & 3) + (e >> 2))];
    }
    __syncthreads();
    double f_pos = 0.0;
    if (wave == 0) {
      const GoalOut go = goal_terms_wave(rb, sp1, goals + (size_t)b * 16, 1, nullptr, s_gaff, s_screw, s_gblk, lane);
      f_pos = go.f_goal;
      if (lane == 0) s_red[0] = f_pos;
    }
    if (collide) {
      const bool need_grad = sp.grad_mode == GTO_GRAD_CENTRAL_DIFF;
      // broad phase (as in k_obstacle_gram): lane i tests the bounding sphere of the wave's i-th chunk against the distance
      // field; a chunk whose sphere cannot reach a non-zero voxel record would add exact zeros to every sum and is skipped
      unsigned long long keepm = 0ull;
      {
        bool keep = false;
        if (lane < c1 - c0) {
          keep = true;
          const double* V = s_vis + ck_link * 12;
          const double u0 = (V[0] * ck_cx + V[1] * ck_cy + V[2] * ck_cz + V[3] + bx - sc.ox) * sc.rinv;
          const double u1 = (V[4] * ck_cx + V[5] * ck_cy + V[6] * ck_cz + V[7] + by - sc.oy) * sc.rinv;
          const double u2 = (V[8] * ck_cx + V[9] * ck_cy + V[10] * ck_cz + V[11] + bz - sc.oz) * sc.rinv;
          const int R = (int)ceil(ck_r * sc.rinv + 1e-6) + GTO_BROAD_MARGIN;
          if (R < GTO_DIST_CAP) {
            const int k0 = min(max((int)floor(u0), 0), sc.nx - 1), k1 = min(max((int)floor(u1), 0), sc.ny - 1), k2 = min(max((int)floor(u2), 0), sc.nz - 1);
            keep = (int)as_global(sc.d_obs)[k2 + sc.nz * (k1 + sc.ny * k0)] <= R;
          }
        }
        keepm = __ballot(keep);
      }
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0;
      int cur_link = -1;
      auto flush = [&]() {
        // four sums per pass (wave_sum4: the same adds in the same order as wave_sum, the results in the four 16-lane rows:
        // row 0 the first, row 2 the second, row 1 the third, row 3 the fourth argument)
        const double t = wave_sum4(a0, a1, a2, a3), u = wave_sum4(a4, a5, a6, 0.0);
        if ((lane & 15) == 0) {
          const int e = ((lane >> 5) & 1) | ((lane >> 3) & 2);  // rows 0, 1, 2, 3 -> entries 0, 2, 1, 3
          double* dst = s_acc + (wave * L + cur_link) * 8;
          dst[e] += t;
          if (e < 3) dst[4 + e] += u;
        }
        a0 = a1 = a2 = a3 = a4 = a5 = a6 = 0.0;
      };
      // GTO_IK_PD surviving chunks at a time, in stages: all their points are requested, then all their voxel records, and
      // the sums are formed chunk by chunk in the same order as one at a time (two dependent round trips per batch, not
      // per chunk).  A slot without a chunk has count 0: its lanes request nothing.
      while (keepm) {
        int lk[GTO_IK_PD], cnt[GTO_IK_PD], st0[GTO_IK_PD];
#pragma unroll
        for (int u = 0; u < GTO_IK_PD; ++u) {
          const int ci_ = keepm ? __builtin_ctzll(keepm) : 0;
          const bool has = keepm != 0ull;
          keepm &= keepm - 1ull;  // (0 stays 0)
          lk[u] = has ? __builtin_amdgcn_readlane(ck_link, ci_) : -1;
          cnt[u] = has ? __builtin_amdgcn_readlane(ck_count, ci_) : 0;
          st0[u] = __builtin_amdgcn_readlane(ck_start, ci_);
        }
        double x0[GTO_IK_PD], x1[GTO_IK_PD], x2[GTO_IK_PD];
#pragma unroll
        for (int u = 0; u < GTO_IK_PD; ++u) {
          const int pi = lane < cnt[u] ? st0[u] + lane : st0[0];  // (a valid address for the idle lanes: no branch around the loads)
          x0[u] = px[pi], x1[u] = py[pi], x2[u] = pz[pi];
        }
        double y0[GTO_IK_PD], y1[GTO_IK_PD], y2[GTO_IK_PD];
        int off[GTO_IK_PD];
#pragma unroll
        for (int u = 0; u < GTO_IK_PD; ++u) {
          const double* V = s_vis + (lk[u] >= 0 ? lk[u] : 0) * 12;
          y0[u] = V[0] * x0[u] + V[1] * x1[u] + V[2] * x2[u] + V[3];
          y1[u] = V[4] * x0[u] + V[5] * x1[u] + V[6] * x2[u] + V[7];
          y2[u] = V[8] * x0[u] + V[9] * x1[u] + V[10] * x2[u] + V[11];
          const int ix = voxel_axis_fast(y0[u], cx, bx, sc.ox, sc.res, sc.rinv, sc.nx);
          const int iy = voxel_axis_fast(y1[u], cy, by, sc.oy, sc.res, sc.rinv, sc.ny);
          const int iz = voxel_axis_fast(y2[u], cz, bz, sc.oz, sc.res, sc.rinv, sc.nz);
          off[u] = iz + sc.nz * (iy + sc.ny * ix);
        }
        double4 rec[GTO_IK_PD];
#pragma unroll
        for (int u = 0; u < GTO_IK_PD; ++u) rec[u] = load_record(sc.r_obs, off[u]);
#pragma unroll
        for (int u = 0; u < GTO_IK_PD; ++u) {
          if (lk[u] < 0) break;  // wave-uniform
          if (lk[u] != cur_link) {
            if (cur_link >= 0) flush();
            cur_link = lk[u];
          }
          if (lane < cnt[u]) {
            const double4 lo4 = rec[u];
            a6 += (double)__builtin_bit_cast(float, (unsigned)__double2loint(lo4.w));
            if (need_grad) {
              const double w0 = lo4.x * sc.inv2r, w1 = lo4.y * sc.inv2r, w2 = lo4.z * sc.inv2r;
              a0 += y1[u] * w2 - y2[u] * w1;
              a1 += y2[u] * w0 - y0[u] * w2;
              a2 += y0[u] * w1 - y1[u] * w0;
              a3 += w0;
              a4 += w1;
              a5 += w2;
            }
          }
        }
      }
      if (cur_link >= 0) flush();
    }
    __syncthreads();
    // ---- fold: normal equations and objective of the trial into slot 1 - slot
    const int ts_ = first ? slot : 1 - slot;
    if (tid < 64) s_A[ts_ * 64 + tid] = s_gblk[BLK_JTJ + tid];
    if (collide) {  // the four waves' sums of every (link, entry), in wave order, by one lane each; then the serial sums over the links read one number
      for (int i = tid; i < L * 8; i += 256) {
        const int l = i >> 3, e = i & 7;
        s_v[i] = ((s_acc[(0 * L + l) * 8 + e] + s_acc[(1 * L + l) * 8 + e]) + s_acc[(2 * L + l) * 8 + e]) + s_acc[(3 * L + l) * 8 + e];
      }
      __syncthreads();
    }
    if (tid >= 64 && tid < 72) {
      const int i = tid - 64;
      double g = 0.0;
      if (i < n && collide) {
        const double* si = s_screw + 6 * i;
        for (int l = 0; l < L; ++l) {
          if (!((s_anc[l] >> i) & 1)) continue;
          const double* v = s_v + 8 * l;
          g += si[0] * v[0] + si[1] * v[1] + si[2] * v[2] + si[3] * v[3] + si[4] * v[4] + si[5] * v[5];
        }
      }
      s_b[ts_ * 8 + i] = (i < n) ? s_gblk[BLK_JTR + i] + 0.5 * sp.w_obstacle * g : 0.0;
    }
    if (tid == 72) {
      double csum = 0.0;
      if (collide)
        for (int l = 0; l < L; ++l) csum += s_v[8 * l + 6];
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
      const int bad = gj_invert8(S, r, c, n);
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
// Phase stamps of k_base_solve (builds with -DGTO_DEBUG_BASE_TIMING only: tools/base_stamps.py; the production kernel carries
// none of this): ticks of goal set 0's workgroup summed over its iterations -- [0] kinematics of the goals, [1] goal terms,
// moments and block entries, [2] objective, accept / reject, [3] base gradient and active set, [4] elimination of the goal
// blocks (8 x 8 inverses), [5] 3 x 3 Schur complement and base step, [6] back substitution, projection, predicted decrease,
// [7] evaluations, [8] passes of eight goals.
#ifdef GTO_DEBUG_BASE_TIMING
__device__ long long g_base_dbg[16];
#define GTO_BASE_STAMP(i)                                   \
  do {                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0) {              \
      const long long t_ = clock64();                       \
      g_base_dbg[i] += t_ - t_last_;                        \
      t_last_ = t_;                                         \
    }                                                       \
  } while (0)
#define GTO_BASE_COUNT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_base_dbg[i] += 1; } while (0)
#else
#define GTO_BASE_STAMP(i) do { } while (0)
#define GTO_BASE_COUNT(i) do { } while (0)
#endif
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
#ifdef GTO_DEBUG_BASE_TIMING
  long long t_last_ = clock64();
#endif
  for (;; ++k) {
    GTO_BASE_COUNT(7);
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
      GTO_BASE_STAMP(0);
      GTO_BASE_COUNT(8);
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
      GTO_BASE_STAMP(1);
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
    GTO_BASE_STAMP(2);
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
    GTO_BASE_STAMP(3);
    // (2) eliminate the goal blocks: E_i = D_i^-1 C_i, u_i = D_i^-1 rhs_i, and their Schur contributions.  A wave takes
    // its goals three at a time, side by side: the Gauss-Jordan inversion is a chain of dependent pivots (element by readlane,
    // column by DPP, row by a lane permute, reciprocal, update: ~350 ticks each), and three independent chains in one
    // instruction stream fill each other's waits (one after the other: 19 K ticks of an 83 K evaluation).  Same operations
    // on the same operands per goal: same bits.
    for (int i0 = wave; i0 < ng; i0 += 12) {
      double S[3], C0[3], C1[3], C2[3], bcv[3];
      bool acv[3];
      int bad3 = 0;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = min(i0 + 4 * u, ng - 1);  // (a goal past the end repeats the last one: computed, not stored)
        const double* sys = sysc + (size_t)i * GTO_BASE_SYS;
        const double* xi = s_x + 8 + 8 * i;
        const double xr = xi[r], xc = xi[c], br = sys[100 + r], bc = sys[100 + c];
        const int rr = r < n ? r : 0, cc = c < n ? c : 0;
        const bool ar = r >= n || (xr <= rb->lower[rr] && br > 0.0) || (xr >= rb->upper[rr] && br < 0.0);
        const bool ac = c >= n || (xc <= rb->lower[cc] && bc > 0.0) || (xc >= rb->upper[cc] && bc < 0.0);
        double Sv = sys[lane];
        if (ar || ac) Sv = (r == c) ? 1.0 : 0.0;
        else if (r == c) Sv *= (1.0 + lambda);
        S[u] = Sv;
        acv[u] = ac;
        bcv[u] = bc;
        C0[u] = (ac || acty0) ? 0.0 : sys[64 + 3 * c];
        C1[u] = (ac || acty1) ? 0.0 : sys[64 + 3 * c + 1];
        C2[u] = (ac || acty2) ? 0.0 : sys[64 + 3 * c + 2];
      }
      {  // pivot J of all three blocks before pivot J + 1 of any
        int b0 = 0, b1 = 0, b2 = 0;
#define GTO_PIV3(J) do { gj_pivot8<J>(S[0], b0, r, c); gj_pivot8<J>(S[1], b1, r, c); gj_pivot8<J>(S[2], b2, r, c); } while (0)
        GTO_PIV3(0);
        if (n > 1) GTO_PIV3(1);
        if (n > 2) GTO_PIV3(2);
        if (n > 3) GTO_PIV3(3);
        if (n > 4) GTO_PIV3(4);
        if (n > 5) GTO_PIV3(5);
        if (n > 6) GTO_PIV3(6);
        if (n > 7) GTO_PIV3(7);
#undef GTO_PIV3
        bad3 = b0 | (i0 + 4 < ng ? b1 : 0) | (i0 + 8 < ng ? b2 : 0);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = i0 + 4 * u;
        const double e0 = matvec8(S[u], C0[u]), e1 = matvec8(S[u], C1[u]), e2 = matvec8(S[u], C2[u]), uu = matvec8(S[u], acv[u] ? 0.0 : -bcv[u]);
        if (c == 0 && i < ng) {
          s_E[i * 24 + 3 * r] = e0, s_E[i * 24 + 3 * r + 1] = e1, s_E[i * 24 + 3 * r + 2] = e2;
          s_u[i * 8 + r] = uu;
        }
      }
      if (__any(bad3) && lane == 0) s_red[1] = 1.0;
      wave_sync();
      if (lane < 36) {  // twelve lanes per goal: C^T E (9) and C^T u (3)
        const int u = lane / 12, e = lane - 12 * u, i = i0 + 4 * u;
        if (i < ng) {
          const double* sys = sysc + (size_t)i * GTO_BASE_SYS;
          const double* xi = s_x + 8 + 8 * i;
          const int a = e < 9 ? e / 3 : e - 9, a2 = e < 9 ? e % 3 : -1;
          const bool acta = a == 0 ? acty0 : (a == 1 ? acty1 : acty2);
          double v = 0.0;
          for (int j = 0; j < n; ++j) {
            const double xj = xi[j], bj = sys[100 + j];
            const bool aj = (xj <= rb->lower[j] && bj > 0.0) || (xj >= rb->upper[j] && bj < 0.0);
            const double cj = (aj || acta) ? 0.0 : sys[64 + 3 * j + a];
            v += cj * (a2 >= 0 ? s_E[i * 24 + 3 * j + a2] : s_u[i * 8 + j]);
          }
          s_part[i * 16 + e] = v;
        }
      }
    }
    __syncthreads();
    GTO_BASE_STAMP(4);
    // (3) 3x3 Schur complement in goal order, Cholesky, base step.  Twelve lanes of wave 0 sum one entry each over the goals
    // (the same sums in the same order as one thread after the other: 10 K ticks of dependent LDS reads on one lane before),
    // lane 0 collects them and factorises
    if (tid < 64) {
      const bool act[3] = {acty0, acty1, acty2};
      double mine = 0.0;
      if (lane < 9) {
        const int a = lane / 3, a2 = lane - 3 * a;
        double v = (a == a2) ? w_effort : 0.0;
        for (int i = 0; i < ng; ++i) v += sysc[(size_t)i * GTO_BASE_SYS + 88 + lane];
        if (act[a] || act[a2]) v = (a == a2) ? 1.0 : 0.0;
        else if (a == a2) v *= (1.0 + lambda);
        for (int i = 0; i < ng; ++i) v -= s_part[i * 16 + lane];
        mine = v;
      } else if (lane < 12) {
        const int a = lane - 9;
        double v = act[a] ? 0.0 : -s_red[4 + a];
        for (int i = 0; i < ng; ++i) v -= s_part[i * 16 + lane];
        mine = v;
      }
      double M[9], rh[3];
#pragma unroll
      for (int e = 0; e < 9; ++e) M[e] = __shfl(mine, e, 64);
#pragma unroll
      for (int a = 0; a < 3; ++a) rh[a] = __shfl(mine, 9 + a, 64);
      if (lane == 0) {
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
    }
    __syncthreads();
    if (s_red[1] != 0.0) {
      status = GTO_STATUS_NUMERICAL;
      break;
    }
    GTO_BASE_STAMP(5);
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
    GTO_BASE_STAMP(6);
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

// ---- 16 x 16 blocks held by ONE wave, four entries per lane: lane l = 4 r + g owns row r, columns 4 g .. 4 g + 3
// (k_lm_step_wide).  Sum over the four lanes of a quad, result in all four.
__device__ __forceinline__ double quad_sum4(double v) {
  v = dpp_add<0xB1>(v);  // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// every lane <- v of lane 4 J + (lane & 3): row J's entry of this lane's column group (ds_bpermute; byte index row_src = 4 (lane & 3))
template <int J>
__device__ __forceinline__ double row_bcast16(double v, int row_src) {
  const int lo = __builtin_amdgcn_ds_bpermute(row_src + 16 * J, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(row_src + 16 * J, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// Pivots J and J + 1 (J even) of the in-place Gauss-Jordan inverse (no pivoting: SPD).  The same operations on the same
// operands as one pivot after the other; what changes is when the data moves.  A pivot needs its row in every lane, and
// that row is only final once the pivot before it has been applied: fetched then, the cross-lane trip (ds_bpermute, ~120
// cycles) sits on the dependent chain of every pivot, next to the reciprocal (~100).  Here rows J and J + 1 are both fetched
// BEFORE pivot J (sixteen bpermutes in flight together, beside the first reciprocal), and every lane applies pivot J to its
// copy of row J + 1 itself (the four fused multiply-adds the lanes of that row do anyway); the scalars of the 2 x 2
// pivot block come through readlane up front, so the second pivot element is a fused multiply-add behind the first
// reciprocal, not a trip through the lanes.  Pivot element S[J][J] by readlane, a row's entry of a pivot column by a
// broadcast inside the quad (DPP).
template <int J>
__device__ __forceinline__ void gj_pivot_pair16(double (&S)[4], int& bad, int r, int g, int row_src) {
  static_assert((J & 1) == 0 && (J >> 2) == ((J + 1) >> 2), "an even pivot and its successor share a column group");
  constexpr int JG = J >> 2, JK = J & 3, NK = JK + 1;
  // the 2 x 2 pivot block of the matrix as it is before pivot J (uniform values)
  const double pjj = readlane_f64(S[JK], 4 * J + JG);        // S[J][J]
  const double pjn = readlane_f64(S[NK], 4 * J + JG);        // S[J][J+1]
  const double pnj = readlane_f64(S[JK], 4 * (J + 1) + JG);  // S[J+1][J]
  const double pnn = readlane_f64(S[NK], 4 * (J + 1) + JG);  // S[J+1][J+1]
  double pj[4], pn[4];  // rows J and J + 1, this lane's four columns
#pragma unroll
  for (int k = 0; k < 4; ++k) pj[k] = row_bcast16<J>(S[k], row_src);
#pragma unroll
  for (int k = 0; k < 4; ++k) pn[k] = row_bcast16<J + 1>(S[k], row_src);
  // ---- pivot J
  if (!(pjj > 0.0)) bad = 1;
  const double piv = fast_rcp(pjj);
  {
    const double prj = quad_perm<JG, JG, JG, JG>(S[JK]);  // S[r][J]
    const double tcol = -prj * piv;
    const bool prow = r == J;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool pcol = k == JK && g == JG;
      const double in_row = pcol ? piv : pj[k] * piv;
      const double off_row = pcol ? tcol : fma(tcol, pj[k], S[k]);
      S[k] = prow ? in_row : off_row;
    }
  }
  // row J + 1 after pivot J (what its lanes have just computed, formed here from the copies) and its pivot element
  const double tn = -pnj * piv;
#pragma unroll
  for (int k = 0; k < 4; ++k) pn[k] = (k == JK && g == JG) ? tn : fma(tn, pj[k], pn[k]);
  const double pnn1 = fma(tn, pjn, pnn);
  // ---- pivot J + 1
  if (!(pnn1 > 0.0)) bad = 1;
  const double piv1 = fast_rcp(pnn1);
  {
    const double prj = quad_perm<JG, JG, JG, JG>(S[NK]);  // S[r][J+1], after pivot J
    const double tcol = -prj * piv1;
    const bool prow = r == J + 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool pcol = k == NK && g == JG;
      const double in_row = pcol ? piv1 : pn[k] * piv1;
      const double off_row = pcol ? tcol : fma(tcol, pn[k], S[k]);
      S[k] = prow ? in_row : off_row;
    }
  }
}
// n: optimised joints.  Rows and columns n .. 15 of a block are padding: rows of the identity that nothing couples to,
// and a pivot on one of them changes nothing (pivot 1, a column of zeros, the row itself), so those pivots are not run
// (ten optimised joints: five pairs instead of eight; same values).
__device__ __forceinline__ int gj_invert16(double (&S)[4], int r, int g, int n) {
  int bad = 0;
  const int row_src = 4 * g;  // byte index of lane g (row 0, this lane's column group); + 16 J: lane 4 J + g
  gj_pivot_pair16<0>(S, bad, r, g, row_src);
  if (n > 2) gj_pivot_pair16<2>(S, bad, r, g, row_src);
  if (n > 4) gj_pivot_pair16<4>(S, bad, r, g, row_src);
  if (n > 6) gj_pivot_pair16<6>(S, bad, r, g, row_src);
  if (n > 8) gj_pivot_pair16<8>(S, bad, r, g, row_src);
  if (n > 10) gj_pivot_pair16<10>(S, bad, r, g, row_src);
  if (n > 12) gj_pivot_pair16<12>(S, bad, r, g, row_src);
  if (n > 14) gj_pivot_pair16<14>(S, bad, r, g, row_src);
  return bad;
}

// ------------------------------------------------------------------------------------------------
// Step kernel for robots with 9..16 optimised joints (mobile manipulators: BASELINE configs[4]): the same accept / reject,
// active set, block-tridiagonal solve, projected step and predicted decrease as k_lm_step.  One workgroup of two waves
// (GTO_WIDE_NT threads) per slot.  The serial block recursion is a twisted factorisation: one wave from each end of the
// trajectory, each holding a whole 16 x 16 block, four entries per lane (gj_invert16); the inverses Z_s go to a workspace
// in HBM (L2) and come back for the back substitution.  The data-parallel phases (gradient, active set, projected step,
// predicted decrease) are strided over the workgroup.
//   dynamic LDS (doubles): Q [NP][T] | b, y, x [m][NP] each | e [m][NP] | zd [m][NP] | S [2][NP*NP] | red [32] ; int act [m]
__host__ __device__ inline size_t lm_wide_lds_bytes(int T, int NP) {
  const size_t m = (size_t)T - 2;
  return ((size_t)NP * T + 5 * m * NP + 2 * NP * NP + 32) * sizeof(double) + (m + 8) * sizeof(int);
}

// threads per workgroup of k_lm_step_wide: the two solver waves ARE the workgroup (two more waves would idle through the
// block recursion and hold two of the CU's four SIMDs while they do)
#define GTO_WIDE_NT 128
template <int NP>
__global__ __launch_bounds__(GTO_WIDE_NT) void k_lm_step_wide(const RobotDev* __restrict__ rb, BatchPtrs bp, SolveParams sp, int B,
                                                      double* __restrict__ Zws /* [slots][T-2][NP*NP] */) {
  static_assert(NP == 16, "one thread per entry of a 16x16 block");
  typedef Blk<NP> BK;
  constexpr int NT = GTO_WIDE_NT, NW = NT / 64;
  static_assert(NW == 2 || NW == 4, "two solver waves; the data-parallel phases are strided by the workgroup size");
  const int tid = threadIdx.x, lane = tid & 63;
  if constexpr (GTO_STEP_PRIO != 0) __builtin_amdgcn_s_setprio(GTO_STEP_PRIO);
  if (blockIdx.x == 0 && threadIdx.x == 0 && bp.progress) {  // lagged by design: what had finished when this launch started
    const int nd = __hip_atomic_load(bp.n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(bp.progress, bp.progress_tag | (unsigned long long)(unsigned)nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(bp.progress + 1, bp.progress_tag | (unsigned long long)(unsigned)sp.round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // solve loop only: one workgroup per position of this round's live list; one candidate per round
  const int par = sp.parity, pos_id = blockIdx.x, NS = sp.kcap + 1;
  if (pos_id >= bp.nlive[GTO_NLIVE(par)]) return;
  const int b = bp.live[par * bp.cap + pos_id];
  if (b < 0) return;  // a place left void by a late exit of the last step
  __shared__ int s_nid, s_pos;
  InstState* st = bp.state + b;
  if (st->done) return;
  extern __shared__ __attribute__((aligned(16))) double smem_w[];
  const int T = sp.T, n = rb->n_opt, m = T - 2, nF = rb->n_frames;
  double* s_Q = smem_w;                      // [NP][T]
  double* s_b = s_Q + NP * T;                // [m][NP]
  double* s_y = s_b + m * NP;
  double* s_x = s_y + m * NP;
  double* s_e = s_x + m * NP;
  double* s_zd = s_e + m * NP;               // [m][NP] inverses of the diagonal stretch's blocks (their diagonals)
  double* s_S = s_zd + m * NP;               // [NP*NP] the upward sweep's last inverse, for the meeting block
  double* s_red = s_S + 2 * NP * NP;         // [32]
  int* s_act = reinterpret_cast<int*>(s_red + 32);  // [m] frozen-joint bit masks
  const int r = tid >> 4, c = tid & 15;
  const bool dbg_t = bp.dbg && b == 0 && tid == 0;
  if (dbg_t) bp.dbg[0] = clock64();
  const int trial = (st->slot + 1) % NS;
  double* __restrict__ Zb = Zws + (size_t)pos_id * m * NP * NP;

  // ---- P0: objective of the trial point, from the evaluation records (one per waypoint: sum of c^2, and whether the block
  // holds anything: only those blocks were written, and only those are read below)
  double fo = 0.0;
  unsigned long long nzt0, nzt1;  // non-zero blocks of the trial set: bit s = waypoint s + 2 (nzt0), s + 66 (nzt1)
  {
    static_assert(GTO_MAX_T - 2 <= 128, "two waypoints per lane");
    const double2* __restrict__ wr_ = reinterpret_cast<const double2*>(bp.wrec + ((size_t)trial * B + b) * T * 8);
    const double2 r0_ = 2 + lane < T ? wr_[(size_t)(2 + lane) * 4] : make_double2(0.0, 0.0);
    const double2 r1_ = 66 + lane < T ? wr_[(size_t)(66 + lane) * 4] : make_double2(0.0, 0.0);
    nzt0 = __ballot(r0_.y != 0.0), nzt1 = __ballot(r1_.y != 0.0);
    fo = wave_sum(r0_.x + r1_.x);
    fo += bp.ss_fixed[4 * b] + bp.ss_fixed[4 * b + 1];
  }
  const double f_try = st->fgoal_try[0] + sp.w_obstacle * fo + st->fvel_try[0];
  // ---- P1: accept / reject (uniform over the workgroup)
  double f = st->f, lambda = st->lambda, nu = st->nu;
  int slot = st->slot, done = 0, status = st->status;
  const int k = st->evals;
  const int argmin_try = st->argmin_try[0], argmin_cur0 = st->argmin_cur;
  const double pred0 = st->pred[0];
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
  double* __restrict__ Qt = bp.Qtry + (size_t)b * n * T;  // candidate 0
  if (accept) {
    f = f_try;
    slot = trial;
  }
  const int argmin_cur = accept ? argmin_try : argmin_cur0;
  // non-zero blocks of the set that is current from here on
  const unsigned long long nz0 = accept ? nzt0 : st->nz0, nz1 = accept ? nzt1 : st->nz1;
  auto blk_nz = [&](int s_) -> bool { return s_ < 64 ? (nz0 >> s_) & 1ull : (nz1 >> (s_ - 64)) & 1ull; };
  for (int idx = tid; idx < NP * T; idx += NT) {
    const double v = idx < n * T ? (accept ? Qt : Qc)[idx] : 0.0;
    s_Q[idx] = v;
    if (accept && idx < n * T) Qc[idx] = v;
  }
  if (!done && k >= sp.max_iter) {
    status = GTO_STATUS_MAX_ITER;
    done = 1;
  }
  if (st->first && !(fabs(f) < INFINITY)) {  // a seed whose objective is not a finite number (oracle: solve_instance)
    status = GTO_STATUS_NUMERICAL;
    done = 1;
  }
  int pos_t0 = -1;  // thread 0: this instance's position in the next round's list, once it has one
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
      const int nid_ = atomicAdd(bp.next, 1);     \
      s_nid = nid_ < bp.n_total ? nid_ : -1;      \
      int p_ = pos_t0;                            \
      if (s_nid >= 0 && p_ < 0) {                 \
        p_ = (int)(unsigned)atomicAdd(reinterpret_cast<unsigned long long*>(bp.nlive + GTO_NLIVE(1 - par)), 1ull | (1ull << 32)); /* one candidate per instance: job index = position */ \
      }                                           \
      if (p_ >= 0) {                              \
        bp.live[(1 - par) * bp.cap + p_] = s_nid; \
        bp.jobs[(1 - par) * bp.cap * sp.kcap + p_] = s_nid < 0 ? -1 : GTO_JOB_ENTRY(s_nid, 1, 0); \
      }                                           \
      s_pos = p_;                                 \
    }                                             \
    __syncthreads();                              \
    {                                             \
      const int nid = s_nid;                      \
      if (nid >= 0) {                             \
        double* __restrict__ dst_ = bp.qfs + ((size_t)(1 - par) * bp.cap * sp.kcap + s_pos) * sp.T * nF; \
        for (int i_ = tid; i_ < sp.T * nF; i_ += NT) dst_[i_] = bp.qf[(size_t)nid * sp.T * nF + i_]; \
      }                                           \
    }                                             \
    return;                                       \
  } while (0)
  __syncthreads();
  if (done) GTO_FINISH_W(status);
  if (tid == 0) {  // this instance goes on: its place in the next round's lists (one candidate: job index = position)
    pos_t0 = (int)(unsigned)atomicAdd(reinterpret_cast<unsigned long long*>(bp.nlive + GTO_NLIVE(1 - par)), 1ull | (1ull << 32));
    bp.live[(1 - par) * bp.cap + pos_t0] = b;
    bp.jobs[(1 - par) * bp.cap * sp.kcap + pos_t0] = GTO_JOB_ENTRY(b, (slot + 1) % NS, 0);
    s_pos = pos_t0;
  }

  if (dbg_t) bp.dbg[1] = clock64();
  // ---- P2: gradient b = J^T r at the current iterate, active set, right-hand side
  const double* __restrict__ oblk = bp.blocks + ((size_t)slot * B + b) * T * BK::STRIDE;
  const double* __restrict__ gblk = bp.goalblk + ((size_t)slot * B + b) * 2 * BK::STRIDE;
  const double alpha = sp.alpha;
  for (int s = tid; s < m; s += NT) s_act[s] = 0;
  __syncthreads();
  {
    constexpr int NU2 = ((GTO_MAX_T - 2) * NP + NT - 1) / NT;  // (waypoint, joint) items per thread
    double jv[NU2];  // obstacle J^T r of this thread's items: all requested before the first is used
#pragma unroll
    for (int u = 0; u < NU2; ++u) {
      const int idx = tid + NT * u, i = idx % NP;
      jv[u] = (idx < m * NP && i < n && blk_nz(idx / NP)) ? oblk[(size_t)(idx / NP + 2) * BK::STRIDE + BK::JTR + i] : 0.0;
    }
    const double gj0 = c < n ? gblk[BK::JTR + c] : 0.0, gj1 = (c < n && sp.use_standoff) ? gblk[BK::STRIDE + BK::JTR + c] : 0.0;  // (i == tid & 15 == c for every item)
    const double my_lo = c < n ? rb->lower[c] : 0.0, my_hi = c < n ? rb->upper[c] : 0.0;
#pragma unroll
    for (int u = 0; u < NU2; ++u) {
      const int idx = tid + NT * u;
      if (idx < m * NP) {
        const int sI = idx / NP, i = idx % NP, t = sI + 2;
        double bv = 0.0;
        int act = 1;  // padded rows count as frozen
        if (i < n) {
          bv = sp.w_obstacle * jv[u];
          if (t == T - 1) bv += gj0;
          if (sp.use_standoff && t == sp.ts) bv += gj1;
          const double qt = s_Q[i * T + t], qm = s_Q[i * T + t - 1];
          bv += alpha * (qt - qm);
          if (t < T - 1) bv -= alpha * (s_Q[i * T + t + 1] - qt);
          act = (qt <= my_lo && bv > 0.0) || (qt >= my_hi && bv < 0.0);
        }
        s_b[idx] = bv;
        if (act) atomicOr(&s_act[sI], 1 << i);
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < m * NP; idx += NT) {
    const int sI = idx / NP, i = idx % NP;
    const int a0 = (s_act[sI] >> i) & 1;
    const int a1 = (sI < m - 1) ? (s_act[sI + 1] >> i) & 1 : 1;
    s_e[idx] = (a0 || a1) ? 0.0 : -alpha;
    s_y[idx] = a0 ? 0.0 : -s_b[idx];
  }
  if (tid < 2) s_red[tid] = 0.0;  // failure flags of the two sweeps
  __syncthreads();
  // ---- P3: block-tridiagonal solve by the inverse-based Schur recursion, run from BOTH ends (twisted factorisation, as in
  // k_lm_step): wave 0 eliminates downwards from waypoint 0, wave 1 upwards from the last waypoint, they meet at block `mid`,
  // and the two back-substitutions run outwards from there.  A solver wave holds a whole 16 x 16 block, four entries per
  // lane: lane l = 4 r + g owns row r, columns 4 g .. 4 g + 3.  Gauss-Jordan without pivoting (SPD): the pivot element by
  // readlane, the pivot column by a quad broadcast (DPP), the pivot row by ds_bpermute; no LDS round trip, no barrier
  // inside a sweep (until round 4: one thread per entry, the pivot row and column through LDS, one workgroup barrier
  // per pivot: 1248 barriers and 78 dependent blocks a launch, 377 us).
  //   down: S_s = D_s - E_{s-1} Z_{s-1} E_{s-1}, Z_s = S_s^{-1}, y_s = Z_s (rhs_s - E_{s-1} y_{s-1})
  //   up:   S_s = D_s - E_s Z_{s+1} E_s,         Z_s = S_s^{-1}, y_s = Z_s (rhs_s - E_s y_{s+1})
  //   mid:  S = D - E Z_{mid-1} E - E Z_{mid+1} E, x_mid = S^{-1} (rhs - E y_{mid-1} - E y_{mid+1})
  //   back: x_s = y_s - Z_s E_s x_{s+1} (s < mid),  x_s = y_s - Z_s E_{s-1} x_{s-1} (s > mid)
  // The inverses Z_s go to the workspace in HBM (L2) as [s][lane][4] and come back for the back substitution.
  if (dbg_t) bp.dbg[2] = bp.dbg[3] = clock64();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = lane >> 2, wg = lane & 3;  // row and column group of this lane's four entries
  // Leading diagonal stretch: a waypoint whose obstacle block is zero and that carries no goal term has a DIAGONAL block
  // (velocity term, damping, identity rows of frozen variables), and S stays diagonal until the first dense block: the
  // downward sweep inverts those blocks entry by entry.  The meeting block balances the two chains (a diagonal step costs
  // about a twentieth of a dense block, and every block of the upward sweep is dense: it starts at the goal waypoint).
  int s_dense = nz0 ? __builtin_ctzll(nz0) : (nz1 ? 64 + __builtin_ctzll(nz1) : m);
  s_dense = min(s_dense, sp.use_standoff ? min(sp.ts - 2, T - 3) : T - 3);
  int mid;
  {
    const int sd = s_dense < m ? s_dense : m - 1;
    const int m2 = (20 * (m - 1) + 19 * sd) / 40;  // sd + 20 (mid - sd) = 20 (m - 1 - mid)
    const int m1 = (20 * (m - 1)) / 21;            // mid = 20 (m - 1 - mid)
    mid = m2 >= sd ? m2 : (m1 < sd ? m1 : sd);
    mid = mid < 0 ? 0 : (mid > m - 1 ? m - 1 : mid);
  }
  // undamped entries of waypoint s's block in the wave layout (obstacle + goal + velocity terms) from the obstacle block's
  // four entries o; the goal blocks' entries are loaded once, here
  double g_fin[4], g_so[4];
  {
    const double4 w0 = *reinterpret_cast<const double4*>(gblk + BK::JTJ + 16 * wr + 4 * wg);
    const double4 w1 = sp.use_standoff ? *reinterpret_cast<const double4*>(gblk + BK::STRIDE + BK::JTJ + 16 * wr + 4 * wg) : make_double4(0.0, 0.0, 0.0, 0.0);
    g_fin[0] = w0.x, g_fin[1] = w0.y, g_fin[2] = w0.z, g_fin[3] = w0.w;
    g_so[0] = w1.x, g_so[1] = w1.y, g_so[2] = w1.z, g_so[3] = w1.w;
  }
  // (a block without anything in it was not written: its entries are read from the zeros of the waypoint's record instead,
  // an address select and no branch, so that a batch of these loads is in flight together)
  const double* __restrict__ zeros4 = bp.wrec + (((size_t)slot * B + b) * T + 2) * 8 + 2;
  auto blk4_load = [&](int s) {
    const double* __restrict__ src = blk_nz(s) ? oblk + (size_t)(s + 2) * BK::STRIDE + BK::JTJ + 16 * wr + 4 * wg : zeros4;
    const double2 lo_ = *reinterpret_cast<const double2*>(src), hi_ = *reinterpret_cast<const double2*>(src + 2);
    return make_double4(lo_.x, lo_.y, hi_.x, hi_.y);
  };
  auto blk4_undamped = [&](int s, const double4& o4, double (&a4)[4]) {
    const int t = s + 2;
    const double o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = 4 * wg + k;
      const bool in4 = wr < n && c4 < n;
      double a = in4 ? sp.w_obstacle * o[k] : 0.0;
      if (in4 && wr == c4) a += (t == T - 1) ? alpha : 2.0 * alpha;
      if (in4 && t == T - 1) a += g_fin[k];
      if (in4 && sp.use_standoff && t == sp.ts) a += g_so[k];
      a4[k] = a;
    }
  };
  // ... damped, with the frozen variables' rows and columns replaced by the identity's
  auto blk4_make = [&](int s, const double4& o4, double (&S)[4]) {
    const int am = s_act[s];
    double a4[4];
    blk4_undamped(s, o4, a4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = 4 * wg + k;
      const bool frozen = ((am >> wr) & 1) || ((am >> c4) & 1);
      S[k] = frozen ? (wr == c4 ? 1.0 : 0.0) : (wr == c4 ? a4[k] * (1.0 + lambda) : a4[k]);
    }
  };
  if (wave < 2) {
    int bad = 0;
    const bool down = wave == 0;
    const int s_first = down ? 0 : m - 1, s_stop = mid, dir = down ? 1 : -1;  // s_first, s_first + dir, ... up to (not including) mid
    double Zp[4] = {0.0, 0.0, 0.0, 0.0};
    const bool dbg_w = bp.dbg && b == 0;  // (GTO_DEBUG_TIMING: cycles inside the dense inversions of a sweep, their number, the sweep's cycles)
    long long dbg_gj = 0, dbg_nd = 0;
    const long long dbg_s0 = dbg_w ? clock64() : 0;
    // The downward sweep's diagonal stretch [0, s_dn): a waypoint without an obstacle block and without a goal term has the
    // block 2 alpha (1 + lambda) I (frozen variables: 1), so nothing is loaded and a step is a scalar recurrence per row,
    // every lane its row's: S = d - e^2 z', z = 1 / S, y = z (rhs - e y'); z and y stay in registers from step to step
    int s_begin = s_first;
    if (down) {
      const int s_dn = s_dense < mid ? s_dense : mid;
      const double d_free = wr < n ? 2.0 * alpha * (1.0 + lambda) : 0.0;
      double zp = 0.0, yp = 0.0, smin = 1.0;
      for (int s = 0; s < s_dn; ++s) {
        const bool frozen = (s_act[s] >> wr) & 1;
        const double e1 = s > 0 ? s_e[(s - 1) * NP + wr] : 0.0;
        const double Sd = (frozen ? 1.0 : d_free) - e1 * e1 * zp;
        smin = fmin(smin, Sd);
        zp = fast_rcp(Sd);
        yp = zp * (s_y[s * NP + wr] - e1 * yp);
        if (wg == 0) s_zd[s * NP + wr] = zp, s_x[s * NP + wr] = yp;
      }
      if (!(smin > 0.0) || zp != zp) bad = 1;
      if (s_dn > 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Zp[k] = wr == 4 * wg + k ? zp : 0.0;
      }
      s_begin = s_dn;
      wave_sync();
    }
    double4 on = make_double4(0.0, 0.0, 0.0, 0.0);
    if (s_begin != s_stop) on = blk4_load(s_begin);
    for (int s = s_begin; s != s_stop; s += dir) {
      double S[4];
      const double4 o4 = on;
      if (s + dir != s_stop) on = blk4_load(s + dir);  // the next block's loads fly during this block's inversion
      blk4_make(s, o4, S);
      const double4 rh = *reinterpret_cast<const double4*>(s_y + s * NP + 4 * wg);
      double zc[4] = {rh.x, rh.y, rh.z, rh.w};
      if (s != s_first) {  // (the first dense block of the downward sweep continues the diagonal stretch)
        const int se = down ? s - 1 : s;  // coupling between this block and the previous one of the sweep
        const double er = s_e[se * NP + wr];
        const double4 ec = *reinterpret_cast<const double4*>(s_e + se * NP + 4 * wg);
        const double4 yp = *reinterpret_cast<const double4*>(s_x + (s - dir) * NP + 4 * wg);  // y of the previous block (this wave wrote it)
        S[0] -= er * ec.x * Zp[0], S[1] -= er * ec.y * Zp[1], S[2] -= er * ec.z * Zp[2], S[3] -= er * ec.w * Zp[3];
        zc[0] -= ec.x * yp.x, zc[1] -= ec.y * yp.y, zc[2] -= ec.z * yp.z, zc[3] -= ec.w * yp.w;
      }
      {
        const long long tg0_ = dbg_w ? clock64() : 0;
        bad |= gj_invert16(S, wr, wg, n);
        if (dbg_w) dbg_gj += clock64() - tg0_, ++dbg_nd;
        double2* zdst = reinterpret_cast<double2*>(Zb + ((size_t)s * 64 + lane) * 4);
        zdst[0] = make_double2(S[0], S[1]);
        zdst[1] = make_double2(S[2], S[3]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) Zp[k] = S[k];
      const double yr = quad_sum4(fma(S[3], zc[3], fma(S[2], zc[2], fma(S[1], zc[1], S[0] * zc[0]))));
      if (wg == 0) s_x[s * NP + wr] = yr;  // s_x holds y during the sweeps
      wave_sync();
    }
    if (dbg_w && lane == 0) bp.dbg[56 + 3 * wave] = dbg_gj, bp.dbg[57 + 3 * wave] = dbg_nd, bp.dbg[58 + 3 * wave] = clock64() - dbg_s0;
    if (!down) {  // the upward sweep's last inverse, for the meeting block
      double2* zl = reinterpret_cast<double2*>(s_S + lane * 4);
      zl[0] = make_double2(Zp[0], Zp[1]);
      zl[1] = make_double2(Zp[2], Zp[3]);
    }
    if (__any(bad) && lane == 0) s_red[wave] = 1.0;
    __syncthreads();
    if (down) {  // the meeting block
      double S[4];
      blk4_make(mid, blk4_load(mid), S);
      const double4 rh = *reinterpret_cast<const double4*>(s_y + mid * NP + 4 * wg);
      double zc[4] = {rh.x, rh.y, rh.z, rh.w};
      if (mid > 0) {
        const double er = s_e[(mid - 1) * NP + wr];
        const double4 ec = *reinterpret_cast<const double4*>(s_e + (mid - 1) * NP + 4 * wg);
        const double4 yp = *reinterpret_cast<const double4*>(s_x + (mid - 1) * NP + 4 * wg);
        S[0] -= er * ec.x * Zp[0], S[1] -= er * ec.y * Zp[1], S[2] -= er * ec.z * Zp[2], S[3] -= er * ec.w * Zp[3];
        zc[0] -= ec.x * yp.x, zc[1] -= ec.y * yp.y, zc[2] -= ec.z * yp.z, zc[3] -= ec.w * yp.w;
      }
      if (mid < m - 1) {
        const double er = s_e[mid * NP + wr];
        const double4 ec = *reinterpret_cast<const double4*>(s_e + mid * NP + 4 * wg);
        const double4 yp = *reinterpret_cast<const double4*>(s_x + (mid + 1) * NP + 4 * wg);
        const double4 zu0 = *reinterpret_cast<const double4*>(s_S + lane * 4);
        S[0] -= er * ec.x * zu0.x, S[1] -= er * ec.y * zu0.y, S[2] -= er * ec.z * zu0.z, S[3] -= er * ec.w * zu0.w;
        zc[0] -= ec.x * yp.x, zc[1] -= ec.y * yp.y, zc[2] -= ec.z * yp.z, zc[3] -= ec.w * yp.w;
      }
      bad = gj_invert16(S, wr, wg, n);
      const double xr = quad_sum4(fma(S[3], zc[3], fma(S[2], zc[2], fma(S[1], zc[1], S[0] * zc[0]))));
      if (wg == 0) s_x[mid * NP + wr] = xr;  // x_mid
      if (__any(bad) && lane == 0) s_red[0] = 1.0;
    }
  } else {
    __syncthreads();
  }
  __syncthreads();
  if (s_red[0] != 0.0 || s_red[1] != 0.0) GTO_FINISH_W(GTO_STATUS_NUMERICAL);
  if (dbg_t) bp.dbg[4] = clock64();
  // ---- back substitution outwards from the meeting block, the two waves side by side.  The chain x_mid -> ... -> x_0 is one
  // dependent mat-vec per waypoint, so what counts is the latency of a step: the solution of the previous step stays in
  // registers (its row entry in every lane of the row's quad; its four column entries fetched by ds_bpermute, one trip
  // instead of an LDS write and a read), Z_s E_s is formed off the chain, the four products are summed as a tree, and the
  // inverses come back from the workspace (L2) three steps ahead of their use.  In the diagonal stretch a step is one fused
  // multiply-add per row, its operands in LDS.
  if (wave < 2) {
    const bool down = wave == 0;
    const int dir = down ? -1 : 1;
    const int s_dn = down ? (s_dense < mid ? s_dense : mid) : 0;  // the downward wave's diagonal stretch: [0, s_dn)
    const int s_end = down ? s_dn - 1 : m;                         // dense steps: mid + dir, ..., up to (not including) s_end
    auto zload = [&](int s_) { return *reinterpret_cast<const double4*>(Zb + ((size_t)s_ * 64 + lane) * 4); };
    const double4 zero4 = make_double4(0.0, 0.0, 0.0, 0.0);
    int s = mid + dir;
    double4 z0 = s != s_end ? zload(s) : zero4;
    double4 z1 = (s != s_end && s + dir != s_end) ? zload(s + dir) : zero4;
    double4 z2 = (s != s_end && s + dir != s_end && s + 2 * dir != s_end) ? zload(s + 2 * dir) : zero4;
    double xr = s_x[mid * NP + wr];                                                      // x_{s -+ 1}[r]
    double4 xn = *reinterpret_cast<const double4*>(s_x + mid * NP + 4 * wg);            // x_{s -+ 1}[4 g .. 4 g + 3]
    for (; s != s_end; s += dir) {
      const double4 Z = z0;
      z0 = z1, z1 = z2;
      {
        const int s3 = s + 3 * dir;
        const bool more = down ? s3 > s_end : s3 < s_end;
        z2 = more ? zload(s3) : zero4;
      }
      const int se = down ? s : s - 1;  // coupling between block s and the block nearer to the meeting block
      const double ys = s_x[s * NP + wr];
      const double4 ec = *reinterpret_cast<const double4*>(s_e + se * NP + 4 * wg);
      const double a0 = Z.x * ec.x, a1 = Z.y * ec.y, a2 = Z.z * ec.z, a3 = Z.w * ec.w;
      const double pr = quad_sum4(fma(a1, xn.y, a0 * xn.x) + fma(a3, xn.w, a2 * xn.z));
      xr = ys - pr;
      if (s + dir != s_end) {  // the next step is dense: this solution's entries of the lane's columns
        xn.x = __shfl(xr, 4 * (4 * wg + 0), 64), xn.y = __shfl(xr, 4 * (4 * wg + 1), 64);
        xn.z = __shfl(xr, 4 * (4 * wg + 2), 64), xn.w = __shfl(xr, 4 * (4 * wg + 3), 64);
      }
      if (wg == 0) s_x[s * NP + wr] = xr;
    }
    if (down) {  // the diagonal stretch
      for (s = s_dn - 1; s >= 0; --s) {
        xr = fma(-(s_zd[s * NP + wr] * s_e[s * NP + wr]), xr, s_x[s * NP + wr]);
        if (wg == 0) s_x[s * NP + wr] = xr;
      }
    }
  }
  __syncthreads();
  if (dbg_t) bp.dbg[5] = clock64();
  // ---- P4: projected trial point; s_x becomes the projected step.  The joint values go to the instance's position in
  // the next round's list (full [T][F] table: the frames no optimised joint drives keep the values k_lm_init left in qf)
  double* __restrict__ qfb = bp.qfs + ((size_t)(1 - par) * bp.cap * sp.kcap + s_pos) * T * nF;
  for (int idx = tid; idx < T * nF; idx += NT)
    if (idx < 2 * nF || rb->opt_of_frame[idx % nF] < 0) qfb[idx] = bp.qf[(size_t)b * T * nF + idx];
  double maxstep = 0.0;
  for (int idx = tid; idx < m * NP; idx += NT) {
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
  maxstep = s_red[1];
#pragma unroll
  for (int w_ = 1; w_ < NW; ++w_) maxstep = fmax(maxstep, s_red[1 + w_]);
  if (maxstep < sp.tol_step) GTO_FINISH_W(GTO_STATUS_CONVERGED);
  if (dbg_t) bp.dbg[6] = clock64();
  // ---- P5: predicted decrease of the undamped model: -(2 b.s + s^T A s).  Wave w takes the waypoints s = w (mod NW), a
  // lane four entries of the block (the solve's layout); the blocks' entries are requested several waypoints at a time
  // (one at a time, every waypoint is a dependent trip past this XCD's L2).  The lanes of column group 0 add the terms that
  // are linear in s_r (2 b_r s_r and the coupling -2 alpha s_r s'_r with the next waypoint).
  {
    double part = 0.0;
    constexpr int PB = 8;
    for (int s0 = wave; s0 < m; s0 += NW * PB) {
      double4 o[PB];
#pragma unroll
      for (int u = 0; u < PB; ++u) o[u] = s0 + NW * u < m ? blk4_load(s0 + NW * u) : make_double4(0.0, 0.0, 0.0, 0.0);
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int s = s0 + NW * u;
        if (s < m) {  // wave-uniform
          double a4[4];
          blk4_undamped(s, o[u], a4);
          const double sr = s_x[s * NP + wr];
          const double4 sc4 = *reinterpret_cast<const double4*>(s_x + s * NP + 4 * wg);
          double v = fma(a4[3], sc4.w, fma(a4[2], sc4.z, fma(a4[1], sc4.y, a4[0] * sc4.x)));
          if (wg == 0) {
            const double xn = (s < m - 1) ? s_x[(s + 1) * NP + wr] : 0.0;
            v += 2.0 * fma(-alpha, xn, s_b[s * NP + wr]);
          }
          part = fma(sr, v, part);
        }
      }
    }
    part = wave_sum(part);
    if (lane == 0) s_red[8 + (tid >> 6)] = part;
  }
  __syncthreads();
  double acc = s_red[8];
#pragma unroll
  for (int w_ = 1; w_ < NW; ++w_) acc += s_red[8 + w_];
  if (tid == 0) {
    st->f = f;
    st->lambda = lambda;
    st->nu = nu;
    st->pred[0] = -acc;
    st->ncand = 1;
    st->cflags = 0;
    st->nz0 = nz0, st->nz1 = nz1;
    st->slot = slot;
    st->first = 0;
    st->status = status;
    st->evals = k + 1;
    st->argmin_cur = argmin_cur;
  }
  if (dbg_t) bp.dbg[7] = clock64();
#undef GTO_FINISH_W
}

// ------------------------------------------------------------------------------------------------
// Lanes of one solve call (gto_api.hip): a call's instances are dealt to a few lanes, each with its own stream and lists,
// so that one lane's obstacle launch overlaps another's step launch while the GPU is full.  Towards the end of the call
// every lane is down to a handful of stragglers, and four sparse launch chains sharing the command processor advance at
// 35-55 us a round each where one alone takes 21-27: a lane whose instances are all in flight and few hands them to the
// collecting lane.  This kernel runs on the collector's stream, behind the lane's last step launch (event) and behind the
// collector's: it appends the lane's list of the parity its next round would have read to the collector's list of the
// parity ITS next round reads -- positions, evaluation jobs and the jobs' joint-value rows.  Everything else about an
// instance is indexed by its id in the batch, which the lanes share.  `bound` is the number of instances the host credited
// to the collector (its lagged view of the lane); what had already finished counts as done there.
__global__ __launch_bounds__(256) void k_adopt(BatchPtrs src, int spar, BatchPtrs dst, int dpar, int kcap, int row, int src_total, int bound) {
  __shared__ int s_base, s_map[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = src.nlive[GTO_NLIVE(spar)], nj = src.nlive[GTO_NJOBS(spar)];
  const int p0 = dst.nlive[GTO_NLIVE(dpar)], j0 = dst.nlive[GTO_NJOBS(dpar)];
  int np = 0, nq = 0;
  for (int c = 0; c < n; c += 64) {  // positions: wave 0, void entries dropped
    if (tid < 64) {
      const int id = c + lane < n ? src.live[spar * src.cap + c + lane] : -1;
      const unsigned long long m = __ballot(id >= 0);
      if (id >= 0) dst.live[dpar * dst.cap + min(p0 + np + (int)__popcll(m & ((1ull << lane) - 1ull)), dst.cap - 1)] = id;
      if (lane == 0) s_base = (int)__popcll(m);
    }
    __syncthreads();
    np += s_base;
    __syncthreads();
  }
  for (int c = 0; c < nj; c += 64) {  // jobs, each with its row of joint values
    if (tid < 64) {
      const int e = c + lane < nj ? src.jobs[spar * src.cap * kcap + c + lane] : -1;
      const unsigned long long m = __ballot(e >= 0);
      const int q = min(j0 + nq + (int)__popcll(m & ((1ull << lane) - 1ull)), dst.cap * kcap - 1);
      if (e >= 0) dst.jobs[dpar * dst.cap * kcap + q] = e;
      s_map[lane] = e >= 0 ? q : -1;
      if (lane == 0) s_base = (int)__popcll(m);
    }
    __syncthreads();
    for (int u = 0; u < 64 && c + u < nj; ++u) {
      const int q = s_map[u];
      if (q < 0) continue;
      const double* __restrict__ from = src.qfs + ((size_t)spar * src.cap * kcap + c + u) * row;
      double* __restrict__ to = dst.qfs + ((size_t)dpar * dst.cap * kcap + q) * row;
      for (int i = tid; i < row; i += 256) to[i] = from[i];
    }
    nq += s_base;
    __syncthreads();
  }
  if (tid == 0) {
    dst.nlive[GTO_NLIVE(dpar)] = min(p0 + np, dst.cap);
    dst.nlive[GTO_NJOBS(dpar)] = min(j0 + nq, dst.cap * kcap);
    const int remaining = src_total - __hip_atomic_load(src.n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bound > remaining) atomicAdd(dst.n_done, bound - remaining);
  }
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
// rows of the screw table fk_mfma_tree writes per configuration: one per optimised joint, padded to the block width
__host__ __device__ inline int screw_rows(int n) { return n <= GTO_NB ? GTO_NB : 16; }
__host__ __device__ inline int eval_kin_lds_doubles(int F, int L, int n) {
  return fk_tab_doubles(F, L, n) + GTO_EVAL_TG * F * 2 + fk_scratch_doubles(F, GTO_EVAL_TG) + GTO_EVAL_TG * L * 12 + GTO_EVAL_TG * screw_rows(n) * 6;
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
  fk_mfma_tree(rb, s_tab, ng, s_sc, s_X, reinterpret_cast<int*>(s_X + ng * 32 * F + 64), tid, s_vis, s_screw, nullptr, screw_rows(n));
  __syncthreads();
  if (frames_out) {  // X_f = G_f^T, row-major, in the ping-pong half the last round wrote
    for (int idx = tid; idx < ng * F * 16; idx += 256) {
      const int kq = idx / (F * 16), r_ = idx - kq * F * 16, f = r_ >> 4, e = r_ & 15;
      const double* Xg = s_X + (size_t)kq * 32 * F + (rb->fk_rounds & 1) * 16 * F;
      double v = Xg[fkx(f, 4 * (e & 3) + (e >> 2))];
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
  return fk_tab_doubles(F, L, n) + GTO_PLAN_TG * F * 2 + fk_scratch_doubles(F, GTO_PLAN_TG) + GTO_PLAN_TG * L * 12 + GTO_PLAN_TG * screw_rows(n) * 6 + 4 * GTO_PLAN_TG;
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
  double* s_screw = s_vis + TG * L * 12;            // [TG][screw_rows(n)][6] (by-product of the kinematics, unused here)
  double* s_red = s_screw + TG * screw_rows(rb->n_opt) * 6;  // [4][TG]
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
  fk_mfma_tree(rb, s_tab, ng, s_sc, s_X, reinterpret_cast<int*>(s_X + ng * 32 * F + 64), tid, s_vis, s_screw, nullptr, screw_rows(n));
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
        acc[g] += (double)as_global(sc.c_obs)[iz + sc.nz * (iy + sc.ny * ix)];
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
// mesh_to_sdf/depth_point_cloud.py:126-141 is_outside: the query is in front of the surface the depth image saw at its pixel
// (or projects outside the image)
__device__ __forceinline__ bool depth_is_outside(double q0, double q1, double q2, const float* __restrict__ depth, int H, int W,
                                                 const double* __restrict__ K, const double* __restrict__ cam_inv) {
#pragma clang fp contract(off)
  double pc[3], u[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    pc[r] = ((cam_inv[4 * r] * q0 + cam_inv[4 * r + 1] * q1) + cam_inv[4 * r + 2] * q2) + cam_inv[4 * r + 3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    u[r] = (K[3 * r] * pc[0] + K[3 * r + 1] * pc[1]) + K[3 * r + 2] * pc[2];
  const double ux = u[0] / u[2], uy = u[1] / u[2];
  const bool fx = fabs(ux) < 9.0e18, fy = fabs(uy) < 9.0e18;
  const long ix = fx ? (long)ux : LONG_MIN, iy = fy ? (long)uy : LONG_MIN;
  bool outside = true;
  if (ix >= 0 && iy >= 0 && ix < W && iy < H) outside = pc[2] < (double)depth[iy * W + ix];
  return outside;
}

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

// Voxel centres of a grid given by its three axes (axes = xs | ys | zs), C order, x slowest: the workspace_points of
// gto/gto_models.py:159-165 (numpy.meshgrid(..., indexing="ij") reshaped), generated where they are used
__global__ void k_grid_queries(const double* __restrict__ axes, int nx, int ny, int nz, double* __restrict__ query) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long)nx * ny * nz) return;
  const int iz = (int)(q % nz), iy = (int)((q / nz) % ny), ix = (int)(q / ((long)nz * ny));
  query[3 * q] = axes[ix];
  query[3 * q + 1] = axes[nx + iy];
  query[3 * q + 2] = axes[nx + ny + iz];
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
                                                       float* __restrict__ cost_out, unsigned long long* __restrict__ stats, int cost_only) {
#pragma clang fp contract(off)
  // cost_only (gto_scene_from_depth: only the COST is wanted): a query in front of the surfaces costs nothing once it is
  // epsilon away from every point ((d - epsilon)^2 / (2 epsilon) for 0 < d < epsilon, else 0: depth_point_cloud.py:86-89),
  // so its search starts from the bound epsilon^2 (a hair above: the float comparison `dist < epsilon` must see every
  // point it could) instead of infinity.  A point within epsilon is still found exactly; without one the cost is the same
  // 0; most voxels of a grid with a 0.4 m margin never leave the root.  (The distance written for such queries is the
  // bound, not the distance: sdf_out is not for cost_only callers.)
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
  if (cost_only && live && depth_is_outside(q0, q1, q2, depth, H, W, K, cam_inv)) best = (double)epsilon * (double)epsilon * (1.0 + 1e-6);
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
