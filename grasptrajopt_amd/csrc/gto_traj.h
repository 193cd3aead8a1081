// gto_traj.h — the trajectory solve as ONE kernel: one workgroup per (scene, goal-set) instance runs the whole
// projected Levenberg-Marquardt loop of DESIGN.md section 3 on chip (no launch per iteration, no state in HBM
// between iterations, no host in the loop).
//
//   E phase  (evaluate the trial trajectory)  the waves of the workgroup pull tasks from an LDS counter:
//            a task = a group of G <= 4 consecutive waypoints, handled by ONE wave from start to end without a
//            workgroup barrier:  sin/cos of the joint values -> forward kinematics as a serial walk over the kinematic
//            tree on the FP64 matrix core (v_mfma_f64_4x4x4: the four 4x4x4 products of an instruction are the four
//            waypoints of the group; local transform and chain product per frame, all in registers) -> visual
//            transforms and joint screws to LDS -> broad phase (chunk bounding spheres against the Chebyshev
//            distance field) -> gather loop over the surviving 64-point chunks (one 32-B voxel record per point,
//            wrench lists folded into the 6x6 Gram of the (waypoint, link) key with v_mfma_f64_16x16x4) -> projection
//            of every finished key onto the joint screws, J^T J / J^T r accumulated one entry per lane -> block of a
//            waypoint that touched the obstacle band to the instance's workspace in HBM (L2), its sum of c^2 to LDS.
//            One extra task evaluates the goal-set and velocity terms (FK at the final and the standoff waypoint),
//            and in the first evaluation the pinned waypoints and the links no optimised joint moves are measured once.
//   S phase  (step)  accept / reject, Nielsen damping, active set on the joint bounds, block-tridiagonal solve from
//            both ends (waves 0 and 1), projected trial point, predicted decrease: k_lm_step's arithmetic with the
//            trajectory, the right-hand sides and the factors resident in LDS.
// The two phases share one LDS region.  Results do not depend on which wave takes which task: a (waypoint, link) key
// is folded by one wave in chunk order, the keys of a waypoint in link order.
#pragma once
#include "gto_kernels.h"

#define TRAJ_LIST_CAP 32  // wrench-list entries (8 doubles) per wave

struct TrajArgs {
  // inputs (device)
  const int32_t* scene_id;  // [B]
  const double* qc;         // [B][ndof]
  const double* goals;      // [B][n_max][16]
  const int32_t* n_goals;   // [B]
  const double* standoff;   // [B][16] or null
  const double* base_pos;   // [B][3]
  const double* Q0;         // [B][ndof][T]
  // outputs (device, may be null)
  double* Q_out;            // [B][ndof][T]
  double* dQ_out;           // [B][ndof][T-1]
  double* cost_out;         // [B]
  int32_t* iters_out;       // [B]
  int32_t* status_out;      // [B]
  // workspace: obstacle blocks of the current and the trial iterate, [B][2][T][BLK_STRIDE]; only the blocks of
  // waypoints that touched the obstacle band are ever written or read
  double* blocks;
  // evaluation mode (gto_eval_objective / gto_eval_obstacle_normal_eq): objective terms of Q0 taken as it is
  double* ev_terms;         // [B][4]  f_goal, sum of c^2 (unweighted, all waypoints), f_vel, arg-min goal
  double* ev_blocks;        // [B][T][BLK_STRIDE] or null: J^T J, J^T r, sum c^2 per waypoint (zeros where untouched)
  // work counters of the call: [0] surface points gathered (one voxel record or field value each), [1] chunk
  // bounding spheres tested, [2] trajectory evaluations, [3] instances
  unsigned long long* counters;
  long long* dbg;           // optional phase clocks of instance 0
  int32_t B, raw, eval_only, G;
};

struct TrajLds {  // dynamic LDS layout in doubles, identical on host and device
  int Qc, Qt, qref, margin, ss, ssfix, gaff, goalblk, red, ints, tab, chk, uni;
  int wstride, oV, oScr, oAli, oGk, oSurv, oSsw;  // per-wave scratch of the E phase (offsets inside a wave's slice)
  int Z, y, e, b, x, act;                          // S phase
  int total;
  __host__ __device__ TrajLds(int T, int F, int L, int C, int n_opt, int n_xst, int G, int NW) {
    const int NP = GTO_NB, m = T - 2;
    int o = 0;
    Qc = o;      o += T * NP;
    Qt = o;      o += T * NP;
    qref = o;    o += T * NP;         // configuration at which a waypoint was certified to be in free space
    margin = o;  o += (T + 1) / 2;   // ... and its clearance then, in voxels (int; -1: not certified)
    ss = o;      o += T;
    ssfix = o;   o += 4;
    gaff = o;    o += 48;
    goalblk = o; o += 2 * 2 * BLK_STRIDE;
    red = o;     o += 32;
    ints = o;    o += 16;  // 32 ints
    tab = o;     o += fk_tab_doubles(F, L, n_opt);  // operand table of the kinematics (RobotDev::fk_tab)
    chk = o;     o += 6 * C;                        // chunk table: 48 B per chunk
    uni = o;
    int w = 0;
    oV = w;      w += G * L * 12;
    oScr = w;    w += (G > 2 ? G : 2) * GTO_NB * 6;  // the goal task uses two blocks whatever G is
    oAli = w;    { const int fk = 4 * F * 2 + n_xst * 64, lst = TRAJ_LIST_CAP * 8; w += fk > lst ? fk : lst; }
    oGk = w;     w += 32;
    oSurv = w;   w += 64;
    oSsw = w;    w += 8;
    wstride = w;
    int s = 0;
    Z = s;       s += m * 64;
    y = s;       s += m * NP;
    e = s;       s += m * NP;
    b = s;       s += m * NP;
    x = s;       s += m * NP;
    act = s;     s += (m + 1) / 2 + 1;
    const int e_tot = NW * wstride;
    total = uni + (e_tot > s ? e_tot : s);
  }
};

// what a block (one of the four 4x4x4 products of an FK instruction) of a task works on
struct TrajWp {
  int t;        // waypoint whose configuration is used (-1: idle block)
  int mode;     // 0 regular waypoint (moving links, gradient) | 1 pinned waypoint (all links, value only)
                // | 2 static links under c_all | 3 static links under c_obs | 4 goal-task block (kinematics only)
};
// task kinds: 0 goal terms, 1 fixed (pinned waypoints + static links, first evaluation only), 2 regular group
__device__ __forceinline__ TrajWp traj_wp(int kind, int g, int G, int blk, int T, int ts) {
  TrajWp w;
  w.t = -1;
  w.mode = 0;
  if (kind == 0) {
    if (blk < 2) w.t = blk == 0 ? T - 1 : ts, w.mode = 4;
  } else if (kind == 1) {
    const int v = g * G + blk;
    if (blk < G && v < 4) w.t = v == 1 ? 1 : 0, w.mode = v < 2 ? 1 : v;
  } else {
    const int t = 2 + g * G + blk;
    if (blk < G && t < T) w.t = t;
  }
  return w;
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_traj_solve(
    const RobotDev* __restrict__ rb, const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ pz,
    const Chunk* __restrict__ chunks, const SceneDev* __restrict__ scenes, TrajArgs a, SolveParams sp) {
  extern __shared__ __attribute__((aligned(16))) double smem_tr[];
  constexpr int NP = GTO_NB;
  constexpr int NT = 64 * NW;
  static_assert(NP == 8, "lane-per-entry layout of the 8x8 blocks");
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  // blockIdx -> instance: a contiguous range of instances (one scene's goal sets are neighbours) per XCD
  const int nb8 = (a.B + 7) >> 3;
  const int b = (blockIdx.x & 7) * nb8 + (blockIdx.x >> 3);
  if (b >= a.B) return;
  const int T = sp.T, n = rb->n_opt, ndof = rb->ndof, F = rb->n_frames, L = rb->n_links, C = rb->n_chunks, m = T - 2;
  const int G = a.G;
  const int fr_grip = rb->frame_gripper, fr_ee = rb->frame_ee;
  const TrajLds lay(T, F, L, C, n, rb->n_xst, G, NW);
  double* s_Qc = smem_tr + lay.Qc;
  double* s_Qt = smem_tr + lay.Qt;
  double* s_qref = smem_tr + lay.qref;
  int* s_margin = reinterpret_cast<int*>(smem_tr + lay.margin);
  double* s_ss = smem_tr + lay.ss;
  double* s_ssfix = smem_tr + lay.ssfix;
  double* s_gaff = smem_tr + lay.gaff;
  double* s_goalblk = smem_tr + lay.goalblk;
  double* s_red = smem_tr + lay.red;
  int* s_int = reinterpret_cast<int*>(smem_tr + lay.ints);
  // s_int: [0] task counter | [1] argmin_try | [4..7] touched, slot 0 | [8..11] touched, slot 1 | [12] first dense block
  unsigned* s_touched = reinterpret_cast<unsigned*>(s_int + 4);
  double* s_uni = smem_tr + lay.uni;
  // E phase, this wave's slice
  double* w_base = s_uni + wave * lay.wstride;
  double* s_V = w_base + lay.oV;
  double* s_scr = w_base + lay.oScr;
  double* s_sc = w_base + lay.oAli;            // [4][F][2] sin, cos
  double* s_xst = s_sc + 4 * F * 2;            // [n_xst][64]
  double* s_lst = w_base + lay.oAli;           // wrench list, after the kinematics
  double* s_gk = w_base + lay.oGk;
  unsigned short* s_surv16 = reinterpret_cast<unsigned short*>(w_base + lay.oSurv);  // [256]
  double* s_ssw = w_base + lay.oSsw;
  // S phase
  double* s_Z = s_uni + lay.Z;
  double* s_y = s_uni + lay.y;
  double* s_e = s_uni + lay.e;
  double* s_b = s_uni + lay.b;
  double* s_x = s_uni + lay.x;
  int* s_actm = reinterpret_cast<int*>(s_uni + lay.act);

  const double* __restrict__ Q0b = a.Q0 + (size_t)b * ndof * T;
  double* __restrict__ blk_ws = a.blocks + (size_t)b * 2 * T * BLK_STRIDE;
  const SceneDev sc = scenes[a.scene_id[b]];
  const double bx = a.base_pos[3 * b], by = a.base_pos[3 * b + 1], bz = a.base_pos[3 * b + 2];
  const double cx = (bx - sc.ox) * sc.rinv, cy = (by - sc.oy) * sc.rinv, cz = (bz - sc.oz) * sc.rinv;
  const int nzv = sc.nz;
  const bool dbg = a.dbg && b == 0;

  // ---- seed: optimised rows of Q0, first two waypoints pinned to qc, the rest clipped into the bounds
  // (gto/gto_planner.py:59-72,138); raw: evaluate Q0 as it is
  for (int idx = tid0; idx < T * NP; idx += NT) {
    const int t = idx / NP, j = idx - t * NP;
    double v = 0.0;
    if (j < n) {
      v = Q0b[(size_t)rb->opt_index[j] * T + t];
      if (!a.raw) {
        if (t < 2) v = a.qc[(size_t)b * ndof + rb->opt_index[j]];
        else v = fmin(fmax(v, rb->lower[j]), rb->upper[j]);
      }
    }
    s_Qc[idx] = v;
    s_Qt[idx] = v;
  }
  {
    const int nt = fk_tab_doubles(F, L, n);
    double* s_tab_w = smem_tr + lay.tab;
    for (int i = tid0; i < nt; i += NT) s_tab_w[i] = rb->fk_tab[i];
    double* s_chk_w = smem_tr + lay.chk;
    const double* gch = reinterpret_cast<const double*>(chunks);
    for (int i = tid0; i < 6 * C; i += NT) s_chk_w[i] = gch[i];
  }
  if (tid0 < 16) s_int[tid0] = 0;
  if (tid0 < 4) s_ssfix[tid0] = 0.0;
  for (int t = tid0; t < T; t += NT) s_ss[t] = 0.0, s_margin[t] = -1;
  __syncthreads();

  typedef double gto_v4f64 __attribute__((ext_vector_type(4)));
  const double* __restrict__ tab = smem_tr + lay.tab;
  const Chunk* __restrict__ s_chunks = reinterpret_cast<const Chunk*>(smem_tr + lay.chk);
  // per-frame descriptor, lane f holds frame f's: parent + 1 | (parking slot + 1) << 6 | (link + 1) << 12 |
  // (optimised joint + 1) << 18 | joint type << 23 | (actuated index + 1) << 25; a wave reads it with v_readlane.
  // lane l also holds the ancestor mask of collision link l.
  int fdesc = 0;
  unsigned ldesc = 0u;
  if (lane0 < F)
    fdesc = (rb->parent[lane0] + 1) | ((rb->xst_slot[lane0] + 1) << 6) | ((rb->link_of_frame[lane0] + 1) << 12) |
            ((rb->opt_of_frame[lane0] + 1) << 18) | (rb->joint_type[lane0] << 23) | ((rb->q_index[lane0] + 1) << 25);
  if (lane0 < L) ldesc = rb->link_anc[lane0];
  const double* __restrict__ tVo = tab + GTO_FK_STRIDE * F;
  const double* __restrict__ tU = tVo + 16 * L;
  const bool grad_on = sp.grad_mode == GTO_GRAD_CENTRAL_DIFF;

  // solver state: uniform over the workgroup, every thread carries it
  double f = INFINITY, lambda = sp.lambda0, nu = 2.0, pred = 0.0;
  int first = 1, k = 0, status = GTO_STATUS_MAX_ITER, slot = 0, argmin_cur = 0;
  unsigned long long n_pts = 0, n_tests = 0, n_culled = 0;  // this wave's work counters (lane 0 is authoritative)
  // profiling (a.dbg set): cycles this wave spent in sincos | FK | broad phase | gather | idle at the end of the E phase | S phase
  const bool prof = a.dbg != nullptr;
  long long pc_sc = 0, pc_fk = 0, pc_br = 0, pc_ga = 0, pc_idle = 0, pc_s = 0, pc_t = 0;
  int n_evals = 0;

  for (;;) {
    const int trial = first ? slot : 1 - slot;
    // ================================================================================== E phase
    if (dbg && tid0 == 0) a.dbg[0] = clock64();
    if (tid0 < 4) s_touched[4 * trial + tid0] = 0u;
    __syncthreads();
    {
      const int nGrp = (m + G - 1) / G, nFix = first ? (4 + G - 1) / G : 0;
      const int nTasks = 1 + nFix + nGrp;
      for (;;) {
        int tk = 0;
        if (lane0 == 0) tk = atomicAdd(&s_int[0], 1);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= nTasks) break;
        const int kind = tk == 0 ? 0 : (tk <= nFix ? 1 : 2);
        // regular groups are dealt from the goal end backwards: the waypoints near the goal are the ones near obstacles,
        // their tasks are the long ones and should not start last
        const int g = kind == 0 ? 0 : (kind == 1 ? tk - 1 : nGrp - 1 - (tk - 1 - nFix));
        // lane roles, derived from an opaque copy of the lane id inside every task: otherwise the compiler hoists the
        // per-lane address arithmetic of all phases out of the Levenberg-Marquardt loop and spills hundreds of registers
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int ra = lane >> 4, rc = lane & 3, blk = (lane >> 2) & 3;  // FK: entry (ra, rc) of block blk
        const int fe = 4 * ra + rc, fet = 4 * rc + ra;
        const double ident = (ra == rc) ? 1.0 : 0.0;
        const int r = lane >> 3, c = lane & 7;                           // 8x8 blocks: entry (r, c)
        const int mcol = lane & 15, mrow = lane >> 4;                    // Gram fold: D column, D rows mrow / mrow + 4
        auto gram_index = [](int row, int col) { return col < 6 ? sym6(row, col) : (row < 6 ? 21 + row : -1); };
        const int gk0 = (mrow <= mcol && mcol < 7) ? gram_index(mrow, mcol) : -1;
        const int gk1 = (mrow + 4 <= mcol && mcol < 7) ? gram_index(mrow + 4, mcol) : -1;

        // Temporal culling (exact): a waypoint whose every chunk sphere was at least `margin` voxels clear of any non-zero
        // voxel at configuration qref, and none of whose surface points can have moved further since
        // (|dx| <= sum_j |dq_j| reach_j), still contributes exact zeros: no kinematics, no lookups.
        unsigned cullm = 0u;  // bit blk: block is certified free
        if (kind == 2 && !a.eval_only && rb->reach[0] >= 0.0) {
          const int bq = lane >> 3, jq = lane & 7;
          const TrajWp w = traj_wp(kind, g, G, bq & 3, T, sp.ts);
          const bool in = lane < 32 && w.t >= 0;
          double dv = (in && jq < n) ? fabs(s_Qt[w.t * NP + jq] - s_qref[w.t * NP + jq]) * rb->reach[jq] : 0.0;
          dv = dpp_add<0xB1>(dv);
          dv = dpp_add<0x4E>(dv);
          dv = dpp_add<0x141>(dv);  // sum over the eight lanes of the block
          const int mg = in ? s_margin[w.t] : -1;
          const bool ok = in && mg >= 0 && (int)ceil(dv * sc.rinv) <= mg;
          const unsigned long long bm = __ballot(ok && jq == 0);
          cullm = (unsigned)((bm & 1ull) | ((bm >> 7) & 2ull) | ((bm >> 14) & 4ull) | ((bm >> 21) & 8ull));
          unsigned validm = 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) validm |= traj_wp(kind, g, G, q, T, sp.ts).t >= 0 ? 1u << q : 0u;
          cullm &= validm;
          n_culled += __popc(cullm);
          if (cullm == validm) {  // the whole group is in certified free space
            if (lane < 4 && ((validm >> lane) & 1u)) s_ss[traj_wp(kind, g, G, lane, T, sp.ts).t] = 0.0;
            continue;
          }
        }
        const bool dbg_w = dbg && wave == 0 && kind == 2 && lane == 0;
        if (dbg_w) a.dbg[8] = clock64();
        const long long t_task0 = prof ? clock64() : 0;
        int n_chunks_task = 0;
        // ---- sin / cos of every (block, frame)
        for (int idx = lane; idx < 4 * F; idx += 64) {
          const int bq = idx / F, fq = idx - bq * F;
          const TrajWp w = traj_wp(kind, g, G, bq, T, sp.ts);
          double sn = 0.0, cs = 1.0;
          const int dsc = __shfl(fdesc, fq, 64);
          if (w.t >= 0) {
            const int jt = (dsc >> 23) & 3, dq = ((dsc >> 25) & 63) - 1;
            if (dq >= 0) {
              const int j = ((dsc >> 18) & 31) - 1;
              const double qv = j >= 0 ? s_Qt[w.t * NP + j] : Q0b[(size_t)dq * T + w.t];
              if (jt == GTO_JOINT_REVOLUTE) sincos(qv, &sn, &cs);
              else if (jt == GTO_JOINT_PRISMATIC) sn = qv;
            }
          }
          s_sc[2 * idx] = sn;
          s_sc[2 * idx + 1] = cs;
        }
        if (lane < 4) s_ssw[lane] = 0.0;
        wave_sync();
        if (dbg_w) a.dbg[9] = clock64();
        if (prof) pc_t = clock64(), pc_sc += pc_t - t_task0;
        // ---- forward kinematics: serial walk, one block per waypoint (optas/models.py:826-868).
        // Per frame: L_f = O_f M_f (local; A operand O, B operand M = c0 + cos c1 + sin K), X_f = L_f^T X_parent with the
        // local product already in the A-operand layout of its transpose and X_parent in the layout a D result has.
        const TrajWp myw = traj_wp(kind, g, G, blk, T, sp.ts);
        const bool bvalid = myw.t >= 0;
        {
          double Xprev = ident;
          double Ot = tab[fe], c0 = tab[16 + fe], c1 = tab[32 + fe], Kk = tab[48 + fe];  // frame 0: placement e ^ 0; the table holds the origin transposed
          for (int fq = 0; fq < F; ++fq) {
            const double sn = s_sc[2 * (blk * F + fq)], cs = s_sc[2 * (blk * F + fq) + 1];
            const double Me = fma(sn, Kk, fma(cs, c1, c0));
            const double Lf = __builtin_amdgcn_mfma_f64_4x4x4f64(Ot, Me, 0.0, 0, 0, 0);
            if (fq + 1 < F) {  // operands of the next frame
              const double* kt = tab + GTO_FK_STRIDE * (fq + 1);
              const int fes = fe ^ (5 * ((fq + 1) & 3));  // bank placement of the table entries (gto_device.h, fkx)
              Ot = kt[fes], c0 = kt[16 + fes], c1 = kt[32 + fes], Kk = kt[48 + fes];
            }
            const int dsc = __builtin_amdgcn_readlane(fdesc, fq);
            const int p = (dsc & 63) - 1;
            double Xp = Xprev;
            if (p != fq - 1 || fq == 0) Xp = p < 0 ? ident : s_xst[(((__builtin_amdgcn_readlane(fdesc, p < 0 ? 0 : p) >> 6) & 63) - 1) * 64 + lane];
            const double X = __builtin_amdgcn_mfma_f64_4x4x4f64(Lf, Xp, 0.0, 0, 0, 0);
            const int xs = ((dsc >> 6) & 63) - 1;
            if (xs >= 0) {
              s_xst[xs * 64 + lane] = X;
              wave_sync();
            }
            const int l = ((dsc >> 12) & 63) - 1;
            if (l >= 0 && kind != 0) {  // visual transform V^T = Vo^T X (gto/gto_models.py:92-100)
              const double V = __builtin_amdgcn_mfma_f64_4x4x4f64(tVo[fkx(l, fe)], X, 0.0, 0, 0, 0);
              if (bvalid && blk < G && rc < 3) s_V[(blk * L + l) * 12 + 4 * rc + ra] = V;
            }
            const int j = ((dsc >> 18) & 31) - 1;
            if (j >= 0) {  // world screw of optimised joint j: (a ; o x a), (0 ; a) for a prismatic joint
              const double S = __builtin_amdgcn_mfma_f64_4x4x4f64(tU[fkx(j, fe)], X, 0.0, 0, 0, 0);
              const bool prism = ((dsc >> 23) & 3) == GTO_JOINT_PRISMATIC;
              const double av = S, ov = __shfl(S, (lane + 16) & 63, 64);
              const double a1 = quad_perm<1, 2, 0, 3>(av), a2 = quad_perm<2, 0, 1, 3>(av);
              const double o1 = quad_perm<1, 2, 0, 3>(ov), o2 = quad_perm<2, 0, 1, 3>(ov);
              const double cr = o1 * a2 - o2 * a1;
              if (bvalid && ra == 0 && rc < 3) {
                double* sv = s_scr + (blk * GTO_NB + j) * 6;
                sv[rc] = prism ? 0.0 : av;
                sv[3 + rc] = prism ? av : cr;
              }
            }
            if (kind == 0 && bvalid && rc < 3) {  // gripper and ee frames of the goal task: X = G^T
              if (fq == fr_grip) s_gaff[24 * blk + 4 * rc + ra] = X;
              if (fq == fr_ee) s_gaff[24 * blk + 12 + 4 * rc + ra] = X;
            }
            Xprev = X;
          }
        }
        wave_sync();
        if (dbg_w) a.dbg[10] = clock64();
        if (prof) { const long long t_ = clock64(); pc_fk += t_ - pc_t; pc_t = t_; }
        if (kind == 0) {
          // ---- goal-set terms (gto/gto_planner.py:84-105) and velocity term (:133-135) of the trial
          const GoalOut go = goal_terms_wave(rb, sp, a.goals + (size_t)b * sp.n_max * 16, a.n_goals[b],
                                             a.standoff ? a.standoff + (size_t)b * 16 : nullptr, s_gaff, s_scr,
                                             s_goalblk + trial * 2 * BLK_STRIDE, lane);
          double fv = 0.0;
          for (int idx = lane; idx < n * (T - 2); idx += 64) {
            const int j = idx / (T - 2), t = 1 + idx % (T - 2);
            const double v = (s_Qt[(t + 1) * NP + j] - s_Qt[t * NP + j]) / sp.dt;
            fv += v * v;
          }
          fv = wave_sum(fv);
          if (lane == 0) {
            s_red[16] = go.f_goal;
            s_red[17] = sp.w_vel * fv;
            s_int[1] = go.argmin;
          }
          continue;
        }
        // ---- broad phase + gather, 64 (block, chunk) candidates at a time, in (waypoint, link) order
        gto_v4f64 gD = {0.0, 0.0, 0.0, 0.0};
        double ssl = 0.0;        // sum of c^2 over this lane's points of the current key
        double jtj = 0.0, jtr = 0.0;  // lane (r,c): J^T J entry of the waypoint being accumulated; lanes 0..7: J^T r
        int cnt = 0, cur_key = -1, acc_blk = -1;
        bool acc_touched = false;
        const double inv2r = sc.inv2r;
        auto use_all = [&](const TrajWp& w) { return w.mode == 2 ? true : (w.mode == 3 ? false : w.t < sp.ts); };
        auto drain = [&]() {
          wave_sync();
          for (int e_ = 0; e_ < cnt; e_ += 4) {
            const int ei_ = e_ + mrow;
            const double xv_ = (ei_ < cnt && mcol < 7) ? s_lst[ei_ * 8 + mcol] : 0.0;
            gD = __builtin_amdgcn_mfma_f64_16x16x4f64(xv_, xv_, gD, 0, 0, 0);
          }
          __builtin_amdgcn_wave_barrier();
          cnt = 0;
        };
        auto emit_block = [&]() {  // the waypoint of block acc_blk is complete: its J^T J / J^T r go to the workspace
          if (acc_blk >= 0 && acc_touched) {
            const int t = traj_wp(kind, g, G, acc_blk, T, sp.ts).t;
            double* out = blk_ws + ((size_t)trial * T + t) * BLK_STRIDE;
            out[BLK_JTJ + lane] = jtj;
            if (lane < NP) out[BLK_JTR + lane] = jtr;
            if (lane == 0) atomicOr(&s_touched[4 * trial + (t >> 5)], 1u << (t & 31));
          }
          jtj = 0.0, jtr = 0.0;
          acc_touched = false;
        };
        auto flush = [&](int key) {
          if (cnt) drain();
          const int fb = key >> 16, fl = key & 0xffff;
          const bool w0 = gk0 >= 0, w1 = gk1 >= 0;
          const bool nz = __ballot((w0 && gD[0] != 0.0) || (w1 && gD[1] != 0.0)) != 0ull;
          if (fb != acc_blk) {
            emit_block();
            acc_blk = fb;
          }
          if (nz) {  // wave-uniform: project the key's wrench Gram onto the joint screws
            if (w0) s_gk[gk0] = gD[0];
            if (w1) s_gk[gk1] = gD[1];
            wave_sync();
            const uint32_t anc = __builtin_amdgcn_readlane(ldesc, fl);
            const double* sr = s_scr + (fb * GTO_NB + r) * 6;
            const double* scc = s_scr + (fb * GTO_NB + c) * 6;
            if (r < n && c < n && ((anc >> r) & 1u) && ((anc >> c) & 1u)) {
              double v = 0.0;
#pragma unroll
              for (int p = 0; p < 6; ++p) {
                double u = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) u += s_gk[sym6(p, q)] * scc[q];
                v += sr[p] * u;
              }
              jtj += v;
            }
            if (lane < n && ((anc >> lane) & 1u)) {
              const double* si = s_scr + (fb * GTO_NB + lane) * 6;
              jtr += si[0] * s_gk[21] + si[1] * s_gk[22] + si[2] * s_gk[23] + si[3] * s_gk[24] + si[4] * s_gk[25] + si[5] * s_gk[26];
            }
            acc_touched = true;
            __builtin_amdgcn_wave_barrier();
          }
          gD = gto_v4f64{0.0, 0.0, 0.0, 0.0};
          const double sw = wave_sum(ssl);
          if (lane == 0) s_ssw[fb] += sw;
          ssl = 0.0;
        };
        struct ChunkLite {
          int key, start, count;
        };
        struct Staged {
          int key, count;
          double y0, y1, y2;
          double4 rec;
          float fval;
        };
        // Survivors of a super-batch: s_surv16[i] = chunk | block << 8.  Gather loop over them as ONE software-pipelined
        // loop with a single copy of every stage: iteration i requests the point coordinates of chunk i + 2, transforms
        // chunk i + 1 and issues its record gather, and consumes chunk i; the final pass (i == NA, after the last
        // super-batch only) folds the open key and emits the last block.
        auto process_survivors = [&](int NA, bool last) {
          if (NA == 0 && !last) return;
          ChunkLite lch = {0, 0, 0}, ach = {0, 0, 0};  // chunk whose points are in flight / are being transformed
          double n0 = 0.0, n1 = 0.0, n2 = 0.0;
          Staged cur = {}, nxt = {};
#pragma unroll 1
          for (int i = -2; i < NA + (last ? 1 : 0); ++i) {
            // ---- stage A of chunk i + 1 (its coordinates were requested one iteration ago)
            ach = lch;
            const double x0 = n0, x1 = n1, x2 = n2;
            // ---- stage L: coordinates of chunk i + 2
            if (i + 2 < NA) {
              const int e16 = s_surv16[i + 2];
              const Chunk* cc = s_chunks + (e16 & 0xff);
              lch = {cc->link | ((e16 >> 8) << 16), cc->start, cc->count};
              n0 = n1 = n2 = 0.0;
              if (lane < lch.count) {
                n0 = px[lch.start + lane];
                n1 = py[lch.start + lane];
                n2 = pz[lch.start + lane];
              }
            }
            cur = nxt;
            if (i + 1 >= 0 && i + 1 < NA) {
              const int kb = ach.key >> 16, link = ach.key & 0xffff;
              const TrajWp w = traj_wp(kind, g, G, kb, T, sp.ts);
              const bool pre = use_all(w);  // gto/gto_planner.py:117-131: c_all before the standoff waypoint
              const bool need_grad = grad_on && w.mode == 0;
              const double* V = s_V + (kb * L + link) * 12;
              // point in the robot-base frame (gto/gto_planner.py:114-116); the field frame adds base_position
              const double y0 = V[0] * x0 + V[1] * x1 + V[2] * x2 + V[3];
              const double y1 = V[4] * x0 + V[5] * x1 + V[6] * x2 + V[7];
              const double y2 = V[8] * x0 + V[9] * x1 + V[10] * x2 + V[11];
              const double u0 = fma(y0, sc.rinv, cx), u1 = fma(y1, sc.rinv, cy), u2 = fma(y2, sc.rinv, cz);
              double k0 = floor(u0), k1 = floor(u1), k2 = floor(u2);
              const double edge = fmax(fmax(fabs((u0 - k0) - 0.5), fabs((u1 - k1) - 0.5)), fabs((u2 - k2) - 0.5));
              if (edge > 0.5 - 1e-9) {  // within 1e-9 of a voxel face: the reference's own order decides (gto/gto_models.py:174-187)
                k0 = floor(((y0 + bx) - sc.ox) / sc.res);
                k1 = floor(((y1 + by) - sc.oy) / sc.res);
                k2 = floor(((y2 + bz) - sc.oz) / sc.res);
              }
              const int ix = min(max((int)k0, 0), sc.nx - 1);
              const int iy = min(max((int)k1, 0), sc.ny - 1);
              const int iz = min(max((int)k2, 0), sc.nz - 1);
              const int off = iz + nzv * (iy + sc.ny * ix);
              nxt.key = ach.key;
              nxt.count = ach.count;
              nxt.y0 = y0, nxt.y1 = y1, nxt.y2 = y2;
              nxt.fval = 0.f;
              nxt.rec = make_double4(0.0, 0.0, 0.0, 0.0);
              if (!need_grad) nxt.fval = (pre ? sc.c_all : sc.c_obs)[off];
              else nxt.rec = *reinterpret_cast<const double4*>(&(pre ? sc.r_all : sc.r_obs)[off]);
            }
            // ---- stage B: consume chunk i
            if (i < 0) continue;
            const int newkey = i < NA ? cur.key : -2;
            if (newkey != cur_key) {
              if (cur_key >= 0) flush(cur_key);
              cur_key = newkey;
            }
            if (i == NA) {
              emit_block();
              break;
            }
            n_pts += cur.count;
            const bool valid = lane < cur.count;
            const TrajWp wk = traj_wp(kind, g, G, cur.key >> 16, T, sp.ts);
            if (!(grad_on && wk.mode == 0)) {
              const double cval = valid ? (double)cur.fval : 0.0;
              ssl = fma(cval, cval, ssl);
            } else {
              const double4 lo4 = cur.rec;
              const double y0 = cur.y0, y1 = cur.y1, y2 = cur.y2;
              const double cval = valid ? (double)__builtin_bit_cast(float, (unsigned)__double2loint(lo4.w)) : 0.0;
              ssl = fma(cval, cval, ssl);
              const double w0 = lo4.x * inv2r, w1 = lo4.y * inv2r, w2 = lo4.z * inv2r;
              const bool act = valid && (w0 != 0.0 || w1 != 0.0 || w2 != 0.0);
              unsigned long long am = __ballot(act);
              while (am) {  // wave-uniform; a list holds TRAJ_LIST_CAP entries
                const int room = TRAJ_LIST_CAP - cnt;
                const int rank = __popcll(am & ((1ull << lane) - 1ull));
                const bool take = act && ((am >> lane) & 1ull) && rank < room;
                if (take) {
                  // wrench of the gradient about the base-frame origin: (y x w, w), then the cost value
                  double2* e = reinterpret_cast<double2*>(s_lst + (cnt + rank) * 8);
                  e[0] = make_double2(y1 * w2 - y2 * w1, y2 * w0 - y0 * w2);
                  e[1] = make_double2(y0 * w1 - y1 * w0, w0);
                  e[2] = make_double2(w1, w2);
                  reinterpret_cast<double*>(e)[6] = cval;
                }
                const unsigned long long tm = __ballot(take);
                cnt += __popcll(tm);
                am &= ~tm;
                if (am || cnt == TRAJ_LIST_CAP) drain();
              }
            }
          }
          wave_sync();  // the survivor list is rewritten by the next super-batch
        };
        // Broad phase, 256 (block, chunk) candidates at a time in (waypoint, link) order: bounding sphere of the chunk against
        // the Chebyshev distance to the nearest non-zero voxel; a chunk that cannot reach one contributes exact zeros and is
        // skipped.  The four distance lookups of a lane are issued together.
        const int nSB = (4 * C + 255) >> 8;
        int slk0 = 1 << 20, slk1 = 1 << 20, slk2 = 1 << 20, slk3 = 1 << 20;  // per block: min over this lane's chunks of (distance - radius)
        int nsv0 = 0, nsv1 = 0, nsv2 = 0, nsv3 = 0;                          // per block: surviving chunks (uniform)
#pragma unroll 1
        for (int sb = 0; sb < nSB; ++sb) {
          int dd[4], Rr[4], e16[4];
          bool pre_keep[4], vld[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int gi = sb * 256 + 64 * u + lane;
            const int kq = gi / C, ci = gi - kq * C;
            const TrajWp w = traj_wp(kind, g, G, kq < 4 ? kq : 3, T, sp.ts);
            vld[u] = kq < 4 && w.t >= 0 && !((cullm >> kq) & 1u);
            pre_keep[u] = false;
            dd[u] = 0, Rr[u] = 0;
            e16[u] = ci | (kq << 8);
            if (vld[u]) {
              const Chunk cc = s_chunks[ci];
              const bool is_static = cc.pad != 0;
              // regular waypoints skip the static links (measured once); the static-only blocks skip everything else
              pre_keep[u] = w.mode == 0 ? !is_static : (w.mode == 1 ? true : is_static);
              if (pre_keep[u]) {
                const double* V = s_V + (kq * L + cc.link) * 12;
                const double u0 = (V[0] * cc.cx + V[1] * cc.cy + V[2] * cc.cz + V[3] + bx - sc.ox) * sc.rinv;
                const double u1 = (V[4] * cc.cx + V[5] * cc.cy + V[6] * cc.cz + V[7] + by - sc.oy) * sc.rinv;
                const double u2 = (V[8] * cc.cx + V[9] * cc.cy + V[10] * cc.cz + V[11] + bz - sc.oz) * sc.rinv;
                const int R = (int)ceil(cc.r * sc.rinv + 1e-6) + GTO_BROAD_MARGIN;
                const int k0 = (int)floor(u0), k1 = (int)floor(u1), k2 = (int)floor(u2);
                Rr[u] = R;
                int sl = -1000;
                // only spheres inside the grid (no clipped indices) and closer than the cap can be culled
                if (R < GTO_DIST_CAP && k0 - R >= 0 && k1 - R >= 0 && k2 - R >= 0 && k0 + R < sc.nx && k1 + R < sc.ny && k2 + R < sc.nz) {
                  const uint8_t* __restrict__ dist = use_all(w) ? sc.d_all : sc.d_obs;
                  dd[u] = (int)dist[k2 + nzv * (k1 + sc.ny * k0)];
                  // clearance that also survives a move of the sphere: stay inside the grid and below the cap
                  sl = min(min(k0, k1), k2) - R;
                  sl = min(sl, min(min(sc.nx - 1 - k0, sc.ny - 1 - k1), sc.nz - 1 - k2) - R);
                  sl = min(sl, GTO_DIST_CAP - 1 - R);
                  sl = min(sl, dd[u] - R);
                }
                slk0 = kq == 0 ? min(slk0, sl) : slk0;
                slk1 = kq == 1 ? min(slk1, sl) : slk1;
                slk2 = kq == 2 ? min(slk2, sl) : slk2;
                slk3 = kq == 3 ? min(slk3, sl) : slk3;
              }
            }
          }
          int NA = 0;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            n_tests += __popcll(__ballot(vld[u]));
            const bool keep = pre_keep[u] && dd[u] <= Rr[u];
            const unsigned long long bm = __ballot(keep);
            if (keep) s_surv16[NA + __popcll(bm & ((1ull << lane) - 1ull))] = (unsigned short)e16[u];
            NA += __popcll(bm);
            if (bm) {
              const int kqu = e16[u] >> 8;
              nsv0 += __popcll(__ballot(keep && kqu == 0));
              nsv1 += __popcll(__ballot(keep && kqu == 1));
              nsv2 += __popcll(__ballot(keep && kqu == 2));
              nsv3 += __popcll(__ballot(keep && kqu == 3));
            }
          }
          wave_sync();
          n_chunks_task += NA;
          if (prof) { const long long t_ = clock64(); pc_br += t_ - pc_t; pc_t = t_; }
          process_survivors(NA, sb == nSB - 1);
          if (prof) { const long long t_ = clock64(); pc_ga += t_ - pc_t; pc_t = t_; }
        }
        if (kind == 2) {  // remember how much room every waypoint of the group had (only if the whole waypoint was culled)
          const int m0 = wave_min_i32(slk0), m1 = wave_min_i32(slk1), m2 = wave_min_i32(slk2), m3 = wave_min_i32(slk3);
          const int bq = lane >> 3, jq = lane & 7;
          const TrajWp w = traj_wp(kind, g, G, bq & 3, T, sp.ts);
          if (lane < 32 && w.t >= 0 && !((cullm >> bq) & 1u)) {
            const int mall = bq == 0 ? m0 : (bq == 1 ? m1 : (bq == 2 ? m2 : m3));
            const int nsv = bq == 0 ? nsv0 : (bq == 1 ? nsv1 : (bq == 2 ? nsv2 : nsv3));
            s_qref[w.t * NP + jq] = s_Qt[w.t * NP + jq];
            if (jq == 0) s_margin[w.t] = (nsv == 0 && mall >= 2) ? mall - 2 : -1;
            if (prof && jq == 0) {
              atomicAdd(a.counters + 11, (unsigned long long)(nsv == 0 && mall >= 2));
              atomicAdd(a.counters + 12, (unsigned long long)(nsv == 0));
              atomicAdd(a.counters + 13, (unsigned long long)(mall >= 2));
              atomicAdd(a.counters + 14, 1ull);
            }
          }
        }
        if (dbg && lane == 0) {  // the longest task of this evaluation, and how many chunks it gathered
          const int dtc = (int)(clock64() - t_task0);
          if (atomicMax(&s_int[13], dtc) < dtc) s_int[14] = n_chunks_task | (tk << 16);
          atomicAdd(&s_int[15], n_chunks_task);
        }
        if (dbg_w) a.dbg[11] = a.dbg[12] = clock64();
        // sum of c^2 per waypoint: its keys in link order (+ the static links, measured in the first evaluation)
        if (lane < 4) {
          const TrajWp w = traj_wp(kind, g, G, lane, T, sp.ts);
          if (w.t >= 0) {
            if (kind == 1) s_ssfix[g * G + lane] = s_ssw[lane];
            else s_ss[w.t] = s_ssw[lane];
          }
        }
      }
    }
    if (prof) pc_t = clock64();
    __syncthreads();
    if (prof) { const long long t_ = clock64(); pc_idle += t_ - pc_t; pc_t = t_; }
    ++n_evals;
    if (dbg && tid0 == 0) {
      a.dbg[1] = clock64();
      a.dbg[13] = s_int[13], a.dbg[14] = s_int[14], a.dbg[15] = s_int[15];
      s_int[13] = s_int[14] = s_int[15] = 0;
    }
    // static links: constant per field, added to every regular waypoint (the sums above hold the moving links)
    for (int t = 2 + tid0; t < T; t += NT) s_ss[t] += s_ssfix[t < sp.ts ? 2 : 3];
    if (tid0 == 0) s_int[0] = 0;
    __syncthreads();

    // ================================================================================== S phase
    int lane = lane0, tid = tid0;
    asm volatile("" : "+v"(lane), "+v"(tid));
    const int r = lane >> 3, c = lane & 7;  // 8x8 blocks: entry (r, c)
    // ---- P0: objective of the trial point
    double fo = 0.0;
    for (int t = 2 + lane; t < T; t += 64) fo += s_ss[t];
    fo = wave_sum(fo);
    fo += s_ssfix[0] + s_ssfix[1];
    const double fgoal_try = s_red[16], fvel_try = s_red[17];
    const int argmin_try = s_int[1];
    const double f_try = fgoal_try + sp.w_obstacle * fo + fvel_try;
    if (a.eval_only) {
      if (tid == 0 && a.ev_terms) {
        double* o = a.ev_terms + 4 * (size_t)b;
        o[0] = fgoal_try, o[1] = fo, o[2] = fvel_try, o[3] = (double)argmin_try;
      }
      if (a.ev_blocks) {
        double* eb = a.ev_blocks + (size_t)b * T * BLK_STRIDE;
        for (int idx = tid; idx < T * BLK_STRIDE; idx += NT) {
          const int t = idx / BLK_STRIDE, e = idx - t * BLK_STRIDE;
          const bool tch = t >= 2 && ((s_touched[4 * trial + (t >> 5)] >> (t & 31)) & 1u);
          double v = 0.0;
          if (e == BLK_SS) v = t < 2 ? s_ssfix[t] : s_ss[t];
          else if (e < BLK_SS && tch) v = blk_ws[((size_t)trial * T + t) * BLK_STRIDE + e];
          eb[idx] = v;
        }
      }
      break;
    }
    // ---- P1: accept / reject (block-uniform)
    int done = 0;
    bool accept = false;
    if (first) {
      accept = true;
    } else if (f_try < f && pred > 0.0) {
      accept = true;
      const double df = f - f_try, rho = df / pred;
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
    if (accept) {
      f = f_try;
      slot = trial;
      argmin_cur = argmin_try;
      for (int idx = tid; idx < T * NP; idx += NT) s_Qc[idx] = s_Qt[idx];
    }
    first = 0;
    if (!done && k >= sp.max_iter) {
      status = GTO_STATUS_MAX_ITER;
      done = 1;
    }
    if (done) break;
    // ---- P2: normal equations at the current iterate (A = J^T J, b = J^T r; f = sum r^2)
    const double* __restrict__ oblk = blk_ws + (size_t)slot * T * BLK_STRIDE;
    const double* gblk = s_goalblk + slot * 2 * BLK_STRIDE;
    const unsigned* tmask = s_touched + 4 * slot;
    auto touched = [&](int t) { return (tmask[t >> 5] >> (t & 31)) & 1u; };
    const double alpha = sp.alpha;
    const bool inb = (r < n) && (c < n);
    constexpr int KMAX = (GTO_MAX_T - 2 + NW - 1) / NW;
    constexpr int NU = (8 * GTO_MAX_T + NT - 1) / NT;
    double av[KMAX];  // undamped obstacle J^T J entry (r,c) of this wave's waypoints
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) {
      const int s = wave + NW * kk;
      av[kk] = (inb && s < m && touched(s + 2)) ? oblk[(size_t)(s + 2) * BLK_STRIDE + BLK_JTJ + lane] : 0.0;
    }
    double jv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid + NT * u, i = idx & 7;
      jv[u] = (idx < m * 8 && i < n && touched((idx >> 3) + 2)) ? oblk[(size_t)((idx >> 3) + 2) * BLK_STRIDE + BLK_JTR + i] : 0.0;
    }
    const double gA0 = inb ? gblk[BLK_JTJ + lane] : 0.0;
    const double gA1 = (inb && sp.use_standoff) ? gblk[BLK_STRIDE + BLK_JTJ + lane] : 0.0;
    const double my_lo = (tid & 7) < n ? rb->lower[tid & 7] : 0.0, my_hi = (tid & 7) < n ? rb->upper[tid & 7] : 0.0;
    __syncthreads();  // s_Qc complete; the E-phase scratch is dead: the union region now holds the S-phase arrays
    if (tid == 0) s_int[12] = m;
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
          if (t == T - 1) bv += gblk[BLK_JTR + i];
          if (sp.use_standoff && t == sp.ts) bv += gblk[BLK_STRIDE + BLK_JTR + i];
          const double qt = s_Qc[t * NP + i], qm = s_Qc[(t - 1) * NP + i];
          bv += alpha * (qt - qm);
          if (t < T - 1) bv -= alpha * (s_Qc[(t + 1) * NP + i] - qt);
          // active set: on a bound with the descent direction pointing outward
          act = (qt <= my_lo && bv > 0.0) || (qt >= my_hi && bv < 0.0);
        }
        s_b[idx] = bv;
        actv[u] = act;
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid + NT * u;
      const unsigned long long bal = __ballot(idx < m * 8 && actv[u] != 0);
      if (idx < m * 8 && (idx & 7) == 0) s_actm[idx >> 3] = (int)((bal >> (lane & 56)) & 0xffull);
    }
    __syncthreads();
    const bool diagl = r == c;
    const double dadd = (inb && diagl) ? 2.0 * alpha : 0.0;
    const double idv = diagl ? 1.0 : 0.0, dmul = diagl ? 1.0 + lambda : 1.0;
    const int lane_bits = (1 << r) | (1 << c);
    constexpr unsigned long long kOffDiag = ~0x8040201008040201ull;
    {
      int first_d = m;
      const int s_goal = T - 3, s_stand = sp.use_standoff ? sp.ts - 2 : -1;
#pragma unroll
      for (int kk = 0; kk < KMAX; ++kk) {
        const int s = wave + NW * kk;
        if (s < m) {
          double aa = fma(sp.w_obstacle, av[kk], dadd);
          if (s == s_goal) {
            aa += gA0;
            if (inb && diagl) aa -= alpha;
          }
          if (s == s_stand) aa += gA1;
          const bool frozen = (s_actm[s] & lane_bits) != 0;
          const double v = frozen ? idv : aa * dmul;
          av[kk] = aa;
          s_Z[(size_t)s * 64 + lane] = v;
          if ((__ballot(v != 0.0) & kOffDiag) && s < first_d) first_d = s;
        }
      }
      if (lane == 0 && first_d < m) atomicMin(&s_int[12], first_d);
    }
    for (int idx = tid; idx < m * 8; idx += NT) {
      const int sI = idx >> 3, i = idx & 7;
      const int a0 = (s_actm[sI] >> i) & 1;
      const int a1 = (sI < m - 1) ? (s_actm[sI + 1] >> i) & 1 : 1;
      s_e[idx] = (a0 || a1) ? 0.0 : -alpha;
      s_y[idx] = a0 ? 0.0 : -s_b[idx];
    }
    __syncthreads();
    const int s_dense = s_int[12];
    if (dbg && tid == 0) a.dbg[2] = clock64();
    // ---- P3: block-tridiagonal solve from both ends (twisted factorisation; see k_lm_step)
    int mid;
    {
      const int sd = s_dense < m ? s_dense : m - 1;
      const int m2 = (10 * (m - 1) + 9 * sd) / 20;
      const int m1 = (10 * (m - 1)) / 11;
      mid = m2 >= sd ? m2 : (m1 < sd ? m1 : sd);
      mid = mid < 0 ? 0 : (mid > m - 1 ? m - 1 : mid);
    }
    if (wave == 0) {
      int fail = 0;
      double zp = 0.0, yp = 0.0;
      const int nd = s_dense < mid ? s_dense : mid;
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
      double Zprev = (r == c) ? zp : 0.0;
      double yprev_c = (nd > 0) ? s_x[(nd - 1) * 8 + c] : 0.0;
      for (int s = nd; s < mid; ++s) {
        double S = s_Z[(size_t)s * 64 + lane];
        double zc = s_y[s * 8 + c];
        if (s > 0) {
          const double er = s_e[(s - 1) * 8 + r], ec = s_e[(s - 1) * 8 + c];
          S -= er * ec * Zprev;
          zc -= ec * yprev_c;
        }
        fail |= gj_invert8(S, lane, r, c);
        s_Z[(size_t)s * 64 + lane] = S;
        Zprev = S;
        const double pr = matvec8(S, zc);
        if (c == 0) s_x[s * 8 + r] = pr;
        yprev_c = __shfl(pr, c << 3, 64);
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
        fail |= gj_invert8(S, lane, r, c);
        s_Z[(size_t)s * 64 + lane] = S;
        Zprev = S;
        const double pr = matvec8(S, zc);
        if (c == 0) s_x[s * 8 + r] = pr;
        yprev_c = __shfl(pr, c << 3, 64);
      }
      if (lane == 0) s_red[1] = __any(fail) ? 1.0 : 0.0;
    }
    __syncthreads();
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
      fail |= gj_invert8(S, lane, r, c);
      const double pr = matvec8(S, zc);
      if (c == 0) s_x[mid * 8 + r] = pr;
      if (lane == 0) s_red[2] = __any(fail) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (s_red[2] != 0.0) {
      status = GTO_STATUS_NUMERICAL;
      break;
    }
    if (wave == 0) {  // outwards to waypoint 0
      const int nd = s_dense < mid ? s_dense : mid;
      double xr = s_x[mid * 8 + r], xc = s_x[mid * 8 + c];
      for (int s = mid - 1; s >= nd; --s) {
        const double pr = matvec8(s_Z[(size_t)s * 64 + lane], s_e[s * 8 + c] * xc);
        xr = s_x[s * 8 + r] - pr;
        if (c == 0) s_x[s * 8 + r] = xr;
        xc = __shfl(xr, c << 3, 64);
      }
      if (r == c) {
        for (int s = nd - 1; s >= 0; --s) {
          xr = s_x[s * 8 + r] - s_Z[(size_t)s * 64 + lane] * (s_e[s * 8 + r] * xr);
          s_x[s * 8 + r] = xr;
        }
      }
    } else if (wave == 1) {  // outwards to the last waypoint
      double xc = s_x[mid * 8 + c];
      for (int s = mid + 1; s < m; ++s) {
        const double pr = matvec8(s_Z[(size_t)s * 64 + lane], s_e[(s - 1) * 8 + c] * xc);
        const double xr = s_x[s * 8 + r] - pr;
        if (c == 0) s_x[s * 8 + r] = xr;
        xc = __shfl(xr, c << 3, 64);
      }
    }
    __syncthreads();
    if (dbg && tid == 0) a.dbg[3] = clock64();
    // ---- P4: projected trial point; s_x becomes the projected step
    double maxstep = 0.0;
    for (int idx = tid; idx < m * 8; idx += NT) {
      const int sI = idx >> 3, i = idx & 7, t = sI + 2;
      double sv = 0.0;
      if (i < n) {
        const double q0 = s_Qc[t * NP + i];
        double v = q0 + s_x[idx];
        v = fmin(fmax(v, my_lo), my_hi);
        s_Qt[t * NP + i] = v;
        sv = v - q0;
      }
      s_x[idx] = sv;
      maxstep = fmax(maxstep, fabs(sv));
    }
    if (tid < 2 * NP) s_Qt[tid] = s_Qc[tid];  // the two pinned waypoints
    maxstep = wave_max(maxstep);
    if (lane == 0) s_red[4 + wave] = maxstep;
    __syncthreads();
    maxstep = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) maxstep = fmax(maxstep, s_red[4 + w]);
    if (maxstep < sp.tol_step) {
      status = GTO_STATUS_CONVERGED;
      break;
    }
    // ---- P5: predicted decrease of the undamped model: -(2 b.s + s^T A s), one partial sum per waypoint, the waypoints
    // added up in a fixed order: the value does not depend on how many waves share the waypoints
    {
      const double c0 = (c == 0) ? 2.0 : 0.0;
#pragma unroll
      for (int kk = 0; kk < KMAX; ++kk) {
        const int s = wave + NW * kk;
        if (s < m) {
          const double sr = s_x[s * 8 + r], scv = s_x[s * 8 + c];
          const double xn = (s < m - 1) ? s_x[(s + 1) * 8 + r] : 0.0;
          const double lin = c0 * fma(-alpha, xn, s_b[s * 8 + r]);
          const double pw = wave_sum(sr * fma(av[kk], scv, lin));
          if (lane == 0) s_y[s] = pw;  // s_y is dead after the back-substitution
        }
      }
    }
    __syncthreads();
    {
      double acc = 0.0;
      for (int sI = lane; sI < m; sI += 64) acc += s_y[sI];
      pred = -wave_sum(acc);
    }
    ++k;
    __syncthreads();  // the S-phase arrays are dead: the union region goes back to the waves
    if (prof) pc_s += clock64() - pc_t;
    if (dbg && tid == 0) a.dbg[4] = clock64();
  }

  // ---- results: the current iterate with the parameter rows of Q0 (optas/solver.py:139-157); dQ from Q
  __syncthreads();
  if (!a.eval_only) {
    if (a.Q_out) {
      double* Qo = a.Q_out + (size_t)b * ndof * T;
      for (int idx = tid0; idx < ndof * T; idx += NT) {
        const int dq = idx / T, t = idx - dq * T, j = rb->opt_of_dof[dq];
        Qo[idx] = j >= 0 ? s_Qc[t * NP + j] : Q0b[idx];
      }
    }
    if (a.dQ_out) {
      double* dQo = a.dQ_out + (size_t)b * ndof * (T - 1);
      for (int idx = tid0; idx < ndof * (T - 1); idx += NT) {
        const int dq = idx / (T - 1), t = idx - dq * (T - 1), j = rb->opt_of_dof[dq];
        dQo[idx] = (j >= 0 && t >= 1) ? (s_Qc[(t + 1) * NP + j] - s_Qc[t * NP + j]) / sp.dt : 0.0;
      }
    }
    if (tid0 == 0) {
      if (a.cost_out) a.cost_out[b] = f;
      if (a.iters_out) a.iters_out[b] = k;
      if (a.status_out) a.status_out[b] = status;
    }
  }
  if (a.counters) {
    if (lane0 == 0) {
      atomicAdd(a.counters + 0, n_pts);
      atomicAdd(a.counters + 1, n_tests);
      atomicAdd(a.counters + 10, n_culled);
      if (prof) {
        atomicAdd(a.counters + 4, (unsigned long long)pc_sc);
        atomicAdd(a.counters + 5, (unsigned long long)pc_fk);
        atomicAdd(a.counters + 6, (unsigned long long)pc_br);
        atomicAdd(a.counters + 7, (unsigned long long)pc_ga);
        atomicAdd(a.counters + 8, (unsigned long long)pc_idle);
        atomicAdd(a.counters + 9, (unsigned long long)pc_s);
      }
    }
    if (tid0 == 0) {
      atomicAdd(a.counters + 2, (unsigned long long)n_evals);
      atomicAdd(a.counters + 3, 1ull);
    }
  }
  (void)argmin_cur;
}
