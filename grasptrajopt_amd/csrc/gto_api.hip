// gto_api.hip — host side of libgto_hip.so: the C ABI declared in include/gto_solver.h.
// Owns all device memory behind the opaque handle; no torch, no CPU fallback.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include <sys/prctl.h>

#include "gto_kernels.h"

#define GTO_VERSION GTO_ABI_VERSION  // include/gto_solver.h
#ifndef GTO_OBS_DEEP_PD
#define GTO_OBS_DEEP_PD 8
#endif

static std::string g_create_error;

#define GTO_SWEEP_WGS 64   // workgroups of the crew behind an itemized obstacle launch laid out over an estimate (launch_obstacle)
#define GTO_MAX_LANES 8  // lanes of one solve call (streams, list sets, progress words)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct gto_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;  // false after gto_set_stream(h, caller's stream)
  std::string err;
  gto_solver_opts opts;
  RobotDev rb;  // host copy
  RobotDev* d_rb = nullptr;
  double *d_px = nullptr, *d_py = nullptr, *d_pz = nullptr;
  int32_t *d_plink = nullptr, *d_perm = nullptr;
  Chunk* d_chunks = nullptr;
  PbChunk* d_pbchunks = nullptr;  // bounding spheres of the chunks of moving links, with their frames (prebroad_tail)
  int pb_C = 0;
  std::vector<SceneDev> scenes;  // host mirror, index = scene id
  SceneDev* d_scenes = nullptr;
  size_t d_scenes_cap = 0;
  // solve workspace (grown on demand)
  DevBuf state, Qcur, Qtry, vis, screw, blocks, goalblk, ssfixed, ndone, qf, livebuf, qfs, wrecbuf, itembuf;
  DevBuf counters;  // work counters of a profiled solve
  int mode = GTO_MODE_ROUNDS;  // gto_set_mode: rounds of two launches over the instances in flight (the only mode left)
  unsigned long long last_counters[4] = {0, 0, 0, 0};
  int32_t* h_ndone = nullptr;  // pinned
  // pinned, device-visible, written by the first workgroup of every step launch: word 0 = call tag << 32 | instances
  // finished, word 1 = call tag << 32 | round of that launch.  The host sizes its launches by the first and stays at most
  // `ahead` rounds in front of the second: no event, no copy, nothing in the stream between the kernels
  unsigned long long* h_progress = nullptr;
  unsigned long long* d_progress = nullptr;  // its device address
  unsigned progress_tag = 0;
  int ahead = 8;           // GTO_AHEAD: rounds the host may enqueue beyond the last one it has seen running
  int nap_us = 50, nap_few_us = 10;  // GTO_NAP_US / GTO_NAP_FEW_US: the throttle's naps (the host thread of a lane sleep-polls two pinned words)
  int ahead_few = 4;       // ... in launches with few instances in flight (short rounds: four of them cover the host's launch time, and every round enqueued beyond the last instance's end runs empty)
  // speculation (gto_kernels.h GTO_KSPEC): candidates a step generates ahead of their evaluation in launches with few
  // instances in flight (more work, fewer dependent rounds).  Every candidate is a job of the next obstacle launch, and
  // that launch stays one wave of workgroups up to about `spec_jobs` jobs: with n instances in flight a step hands out
  // up to spec_jobs / n candidates (at most spec_acc after an accepted evaluation, spec_rej after a round without one);
  // beyond that, after a rejection only, spec_rej_few while at most spec_few are in flight.  spec_deep caps n for the
  // candidates after an accepted evaluation.
  int spec_rej = 4, spec_acc = 4, spec_deep = 32, spec_kmax = 1;
  int spec_rej_few = 2, spec_jobs = 20;
  int spec_streak = 4;  // GTO_SPEC_STREAK: first candidate accepted this many rounds in a row -> one candidate after the next accepted evaluation (0: off)
  int obs_deep_max = 32;  // GTO_OBS_DEEP_MAX: ... only up to this many instances in flight (two workgroups per CU: beyond that the five-per-CU variant gets through a launch faster)
  int obs_deep = 1;   // GTO_OBS_DEEP: launches with few instances in flight use the obstacle kernel variant with deep gather batches
  // The tail of a LARGE call (more than spec_deep instances from the start) shares the GPU with the other lanes' launches,
  // and there an evaluation that turns out not to be needed costs more than the round it might save: fewer candidates
  // after an accepted evaluation, the single candidate after a shorter run of first-try accepts, three waypoints per
  // obstacle workgroup (GTO_SPEC_ACC_TAIL, GTO_SPEC_STREAK_TAIL, GTO_OBS_TG_FEW_TAIL; the driver's 20-step call 253 -> 262 k).
  // A call that is small from its first round (one grasp, a batch of a few) is alone on the GPU and keeps the settings
  // that give the shortest chain of rounds (one instance: 0.66 ms; with the tail's settings 0.90 ms).
  int spec_acc_tail = 2, spec_streak_tail = 2, obs_tg_few_tail = 3;
  int spec_few = 64;  // GTO_SPEC_FEW: speculation starts once at most this many instances are in flight: before that the GPU is full and every extra evaluation costs time
  int dbg_cut = 0;
  int dist_relax = 0;  // GTO_DIST_RELAX: build the distance fields by relaxation sweeps instead of the separable passes
  // GTO_OBS_INTERLEAVE: waypoints of an obstacle workgroup nG apart instead of consecutive, so that the waypoints next to
  // the obstacles (neighbours in time) land in different workgroups: 0 never, 1 always, 2 (default) in launches with few
  // instances in flight, where the longest workgroup decides the round (+5 % for one batch at a time, -2 % at saturation)
  int obs_interleave = 2;
  int static_pos = 1;  // GTO_STATIC_POS=0: every instance draws its list positions from the counters in every round
  int pb_merge = 4;  // GTO_PB_MERGE: chunks of a link under one sphere of the step kernel's broad phase
  int prebroad = 1;  // GTO_PREBROAD=0: every (job, group) gets a workgroup of the obstacle kernel in every round
  double pb_min_gain = 0.10;  // GTO_PB_MIN_GAIN: a call whose step-kernel broad phase settles less than this share of the groups stops running it
  int obs_tg_few = 2;  // ... when few instances are in flight (one small batch, the tail of a call): lower latency per round; results do not depend on the group size
  int few_instances = 192;
  int step_nw_few = 8;  // GTO_STEP_NW_FEW: wavefronts per workgroup of the step kernel in launches with few instances in flight (4 or 8)
  int obs_tg = 3;  // waypoints per workgroup of the obstacle kernel: they share the table staging, the FK barriers and the launch overhead
  long long* dbg = nullptr;
  // staging for the host-pointer entry points
  // buffers of the scene that the last gto_set_scene replaced: the next replacement of the same size takes them instead of
  // going through hipMalloc / hipFree (fourteen calls of 0.2-0.3 ms each: most of a small scene's upload time)
  std::vector<std::pair<void*, size_t>> spare;
  DevBuf in[8], out[8];
  // pinned twins of the staging buffers: host arrays are copied through them, so that the transfers are real DMA at a
  // steady rate (a hipMemcpyAsync from pageable memory stages inside the runtime: 1-6 ms of jitter per call with four
  // lanes copying at once) and never depend on what kind of memory the caller's arrays live in
  DevBuf pin_in[8], pin_out[8];
  struct PendingOut { void* host; const void* pin; size_t bytes; };
  std::vector<PendingOut> pending_out;  // device -> pinned copies in flight; finish_out() delivers them after the sync
  // profiling of the dominant kernel
  bool profiling = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_variant;  // kernel variant of launch i (GTO_PROF_*: include/gto_solver.h)
  std::vector<long long> ev_wgs;  // its workgroups
  double last_ms = 0.0;
  int last_launches = 0;
  // per kernel variant of the last profiled solve: milliseconds, launches, workgroups launched, surface points gathered
  double prof_ms[GTO_PROF_VARIANTS] = {0, 0, 0, 0};
  long long prof_launches[GTO_PROF_VARIANTS] = {0, 0, 0, 0}, prof_wgs[GTO_PROF_VARIANTS] = {0, 0, 0, 0};
  unsigned long long prof_points[GTO_PROF_VARIANTS] = {0, 0, 0, 0};
  size_t lm_lds = 0;
  int np = GTO_NB;     // block width of the normal equations: 8 (up to eight optimised joints) or 16
  DevBuf zws;          // k_lm_step_wide: block inverses [slots][T-2][np*np]
  int slots = 512;  // instances in flight per lane of a solve call (GTO_SLOTS); a finished instance hands its slot to the next one
  // lanes of a solve call (gto_solve_batch_device): at most lanes_max, each with at least lane_min instances; a lane with
  // at most adopt_below instances left hands them to lane 0 (0: never).  gto_set_lanes / GTO_LANES, GTO_LANE_MIN, GTO_ADOPT
  int lanes_max = 1, lane_min = 256, adopt_below = 0;
  hipStream_t lane_stream[GTO_MAX_LANES] = {};
  hipEvent_t lane_event[GTO_MAX_LANES] = {};
  hipStream_t user_lane_stream[GTO_MAX_LANES] = {};  // gto_set_lane_streams: the caller's streams for the lanes
  int n_user_lane_streams = 0;
  // items (job, group pairs with something to gather) per evaluation job: the largest ratio the rounds of the last call
  // published (the prior of the next call's launches until its own counts arrive), and of the call that is running
  double items_per_job_prior = 0.0, items_per_job_call = 0.0;
  int item_hint_forced = 0;  // GTO_ITEM_HINT: the estimate itself (tests of the crew)
  int hot_variants = 1;  // GTO_OBS_HOT=0: the solve loop's evaluation launches use the general kernel variants
  int item_grid = 1;  // GTO_ITEM_GRID=0: itemized launches laid out over the upper bound of their item lists
  std::mutex* prof_mu = nullptr;  // set while a call with several lanes (host threads) runs: guards the profiling records
};

#define HIPCHK(h, call)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                  \
      (h)->pending_out.clear(); /* an entry point that fails delivers nothing */                     \
      return GTO_ERR_HIP;                                                                            \
    }                                                                                                \
  } while (0)

static int fail(gto_handle* h, int code, const std::string& msg) {
  if (h) h->pending_out.clear();
  if (h) h->err = msg;
  else g_create_error = msg;
  return code;
}

static int ensure(gto_handle* h, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return GTO_OK;
  if (b.p) HIPCHK(h, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4 + 256;
  HIPCHK(h, hipMalloc(&b.p, want));
  b.cap = want;
  return GTO_OK;
}

extern "C" {

void gto_default_opts(gto_solver_opts* o) {
  o->T = 50;  // gto/gto_planner.py:25
  o->Tmax = 10.0;  // :26
  o->standoff_offset = -10;  // :22
  o->w_obstacle = 10.0;  // :131
  o->w_vel = 0.01;  // :135
  o->max_iter = 100;  // :141
  o->tol_step = 1e-7;
  o->tol_rel_f = 1e-8;  // relative decrease of f below which an accepted step ends the solve (scipy least_squares ftol default)
  o->lambda0 = 1e-3;
  o->grad_mode = GTO_GRAD_CENTRAL_DIFF;
}

int32_t gto_version(void) { return GTO_VERSION; }

const char* gto_last_error(const gto_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int validate_opts(const gto_solver_opts* o, std::string& why) {
  if (o->T < 4) { why = "T must be >= 4"; return 0; }
  if (o->T > GTO_MAX_T) { why = "T must be <= 96 (GTO_MAX_T)"; return 0; }
  if (!(o->Tmax > 0)) { why = "Tmax must be positive"; return 0; }
  if (o->max_iter < 0) { why = "max_iter must be >= 0"; return 0; }
  int ts = o->T + o->standoff_offset;
  if (ts < 2 || ts > o->T - 1) { why = "standoff waypoint T+standoff_offset must lie in [2, T-1]"; return 0; }
  if (o->grad_mode != GTO_GRAD_CENTRAL_DIFF && o->grad_mode != GTO_GRAD_ZERO) { why = "unknown grad_mode"; return 0; }
  if (!(o->lambda0 > 0)) { why = "lambda0 must be positive"; return 0; }
  return 1;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is process-wide per kernel: a handle for a smaller robot or goal set must
// never lower what an earlier handle launches with.  One high-water mark per kernel, only ever raised.
// The attribute belongs to the (function, device) pair of the CURRENT device: the marks are kept per device, and every
// caller has done hipSetDevice(h->device) before.
static hipError_t raise_dynamic_lds(const void* kernel, size_t bytes) {
  struct Mark { int device; const void* kernel; size_t bytes; };
  static std::mutex mu;
  static std::vector<Mark> marks;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  for (auto& m : marks)
    if (m.kernel == kernel && m.device == dev) {
      if (bytes <= m.bytes) return hipSuccess;
      const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e == hipSuccess) m.bytes = bytes;
      return e;
    }
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) marks.push_back({dev, kernel, bytes});
  return e;
}

int gto_create(const gto_robot_desc* d, const gto_solver_opts* opts, int device, gto_handle** out) {
  if (!d || !opts || !out) return fail(nullptr, GTO_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  std::string why;
  if (!validate_opts(opts, why)) return fail(nullptr, GTO_ERR_INVALID_ARG, why);
  if (d->n_frames < 1 || d->n_frames > GTO_MAX_FRAMES) return fail(nullptr, GTO_ERR_UNSUPPORTED, "n_frames out of range (max 32)");
  if (d->n_links < 1 || d->n_links > GTO_MAX_LINKS) return fail(nullptr, GTO_ERR_UNSUPPORTED, "n_links out of range (max 32)");
  if (d->n_opt < 1 || d->n_opt > GTO_MAX_OPT) return fail(nullptr, GTO_ERR_UNSUPPORTED, "n_opt out of range (max 16)");
  if (d->ndof < d->n_opt || d->ndof > GTO_MAX_DOF) return fail(nullptr, GTO_ERR_UNSUPPORTED, "ndof out of range (max 32)");
  if (d->n_points < 1) return fail(nullptr, GTO_ERR_INVALID_ARG, "robot has no surface points");
  if (d->n_gripper_points < 1) return fail(nullptr, GTO_ERR_INVALID_ARG, "robot has no gripper points");
  if (d->frame_ee < 0 || d->frame_ee >= d->n_frames || d->frame_gripper < 0 || d->frame_gripper >= d->n_frames)
    return fail(nullptr, GTO_ERR_INVALID_ARG, "frame_ee / frame_gripper out of range");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, GTO_ERR_NO_DEVICE, "no HIP device visible: the GTO solve path has no CPU fallback");
  if (device >= ndev) return fail(nullptr, GTO_ERR_INVALID_ARG, "device index out of range");

  gto_handle* h = new gto_handle();
  if (device >= 0) {
    if (hipSetDevice(device) != hipSuccess) { delete h; return fail(nullptr, GTO_ERR_HIP, "hipSetDevice failed"); }
    h->device = device;
  } else {
    (void)hipGetDevice(&h->device);
  }
  h->opts = *opts;
  if (const char* e = getenv("GTO_AHEAD")) h->ahead = h->ahead_few = std::max(1, atoi(e));
  if (const char* e = getenv("GTO_NAP_US")) h->nap_us = std::max(1, atoi(e));
  if (const char* e = getenv("GTO_NAP_FEW_US")) h->nap_few_us = std::max(1, atoi(e));
  if (const char* e = getenv("GTO_SPEC_REJ")) h->spec_rej = std::max(1, std::min(GTO_KSPEC, atoi(e)));
  if (const char* e = getenv("GTO_SPEC_REJ_FEW")) h->spec_rej_few = std::max(1, std::min(GTO_KSPEC, atoi(e)));
  if (const char* e = getenv("GTO_SPEC_STREAK")) h->spec_streak = h->spec_streak_tail = std::max(0, atoi(e));
  if (const char* e = getenv("GTO_SPEC_STREAK_TAIL")) h->spec_streak_tail = std::max(0, atoi(e));
  if (const char* e = getenv("GTO_SPEC_JOBS")) h->spec_jobs = std::max(1, atoi(e));
  if (const char* e = getenv("GTO_SPEC_ACC")) h->spec_acc = h->spec_acc_tail = std::max(1, std::min(GTO_KSPEC, atoi(e)));
  if (const char* e = getenv("GTO_SPEC_ACC_TAIL")) h->spec_acc_tail = std::max(1, std::min(GTO_KSPEC, atoi(e)));
  if (const char* e = getenv("GTO_SPEC_DEEP")) h->spec_deep = std::max(0, atoi(e));
  if (const char* e = getenv("GTO_SPEC_FEW")) h->spec_few = std::max(0, atoi(e));
  if (const char* e = getenv("GTO_OBS_DEEP")) h->obs_deep = atoi(e) ? 1 : 0;
  if (const char* e = getenv("GTO_PREBROAD")) h->prebroad = atoi(e) != 0;
  if (const char* e = getenv("GTO_PB_MERGE")) h->pb_merge = std::max(1, atoi(e));
  if (const char* e = getenv("GTO_STATIC_POS")) h->static_pos = atoi(e) != 0;
  if (const char* e = getenv("GTO_PB_MIN_GAIN")) h->pb_min_gain = atof(e);
  if (const char* e = getenv("GTO_OBS_INTERLEAVE")) h->obs_interleave = std::max(0, std::min(2, atoi(e)));
  if (const char* e = getenv("GTO_DIST_RELAX")) h->dist_relax = atoi(e) ? 1 : 0;
  if (const char* e = getenv("GTO_DEBUG_CUT")) h->dbg_cut = atoi(e);
  if (h->dbg_cut) fprintf(stderr, "[gto] WARNING: GTO_DEBUG_CUT=%d cuts the obstacle kernel short: timing experiments only, RESULTS ARE GARBAGE\n", h->dbg_cut);
  if (const char* e = getenv("GTO_ITEM_GRID")) h->item_grid = atoi(e) != 0;
  if (const char* e = getenv("GTO_OBS_HOT")) h->hot_variants = atoi(e) != 0;
  if (const char* e = getenv("GTO_ITEM_HINT")) h->item_hint_forced = std::max(0, atoi(e));
  if (const char* e = getenv("GTO_SLOTS")) h->slots = std::max(1, atoi(e));
  if (const char* e = getenv("GTO_LANES")) h->lanes_max = std::max(1, std::min(GTO_MAX_LANES, atoi(e)));
  if (const char* e = getenv("GTO_LANE_MIN")) h->lane_min = std::max(1, atoi(e));
  if (const char* e = getenv("GTO_ADOPT")) h->adopt_below = std::max(0, atoi(e));
  if (const char* e = getenv("GTO_OBS_TG")) h->obs_tg = h->obs_tg_few = h->obs_tg_few_tail = std::max(1, std::min(GTO_MAX_TG, atoi(e)));
  if (const char* e = getenv("GTO_OBS_TG_FEW")) h->obs_tg_few = h->obs_tg_few_tail = std::max(1, std::min(GTO_MAX_TG, atoi(e)));
  if (const char* e = getenv("GTO_OBS_TG_FEW_TAIL")) h->obs_tg_few_tail = std::max(1, std::min(GTO_MAX_TG, atoi(e)));
  if (const char* e = getenv("GTO_FEW_INSTANCES")) h->few_instances = atoi(e);
  if (const char* e = getenv("GTO_STEP_NW_FEW")) h->step_nw_few = atoi(e) == 8 ? 8 : 4;
  if (getenv("GTO_DEBUG_TIMING")) { (void)hipMalloc((void**)&h->dbg, 256 * sizeof(long long)); (void)hipMemset(h->dbg, 0, 256 * sizeof(long long)); }
  RobotDev& rb = h->rb;
  memset(&rb, 0, sizeof rb);
  rb.n_frames = d->n_frames;
  rb.ndof = d->ndof;
  rb.n_opt = d->n_opt;
  rb.n_links = d->n_links;
  rb.n_points = d->n_points;
  rb.n_gripper_points = d->n_gripper_points;
  rb.frame_ee = d->frame_ee;
  rb.frame_gripper = d->frame_gripper;
  for (int i = 0; i < GTO_MAX_DOF; ++i) rb.opt_of_dof[i] = -1;
  for (int j = 0; j < d->n_opt; ++j) {
    if (d->opt_index[j] < 0 || d->opt_index[j] >= d->ndof) { delete h; return fail(nullptr, GTO_ERR_INVALID_ARG, "opt_index out of range"); }
    rb.opt_index[j] = d->opt_index[j];
    rb.opt_of_dof[d->opt_index[j]] = j;
    rb.lower[j] = d->lower[j];
    rb.upper[j] = d->upper[j];
    if (!(d->lower[j] <= d->upper[j])) { delete h; return fail(nullptr, GTO_ERR_INVALID_ARG, "lower > upper"); }
  }
  for (int i = 0; i < d->n_frames; ++i) {
    int p = d->parent[i];
    if (p >= i || p < -1) { delete h; return fail(nullptr, GTO_ERR_INVALID_ARG, "frames must list parents before children"); }
    int jt = d->joint_type[i];
    if (jt != GTO_JOINT_FIXED && jt != GTO_JOINT_REVOLUTE && jt != GTO_JOINT_PRISMATIC) {
      delete h;
      return fail(nullptr, GTO_ERR_UNSUPPORTED, "joint type not supported (optas/models.py:865-866)");
    }
    if (jt != GTO_JOINT_FIXED && (d->q_index[i] < 0 || d->q_index[i] >= d->ndof)) {
      delete h;
      return fail(nullptr, GTO_ERR_INVALID_ARG, "q_index out of range for an actuated joint");
    }
    rb.parent[i] = p;
    rb.joint_type[i] = jt;
    rb.q_index[i] = (jt == GTO_JOINT_FIXED) ? -1 : d->q_index[i];
    double R[9];
    rpy2r(d->origin_rpy + 3 * i, R);
    rt2aff(R, d->origin_xyz + 3 * i, rb.origin[i]);
    const double* ax = d->axis + 3 * i;
    double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (jt != GTO_JOINT_FIXED && !(nrm > 0)) { delete h; return fail(nullptr, GTO_ERR_INVALID_ARG, "zero joint axis"); }
    for (int k = 0; k < 3; ++k) rb.axis_unit[i][k] = (nrm > 0) ? ax[k] / nrm : 0.0;
    rb.opt_of_frame[i] = -1;
    uint32_t anc = (p >= 0) ? rb.frame_anc[p] : 0u;
    if (rb.q_index[i] >= 0)
      for (int j = 0; j < d->n_opt; ++j)
        if (d->opt_index[j] == rb.q_index[i]) {
          rb.opt_of_frame[i] = j;
          anc |= 1u << j;
        }
    rb.frame_anc[i] = anc;
  }
  {
    int depth[GTO_MAX_FRAMES], maxd = 1;
    for (int i = 0; i < d->n_frames; ++i) {
      depth[i] = rb.parent[i] < 0 ? 0 : depth[rb.parent[i]] + 1;
      maxd = std::max(maxd, depth[i]);
    }
    // a frame at depth d is the product of d + 1 local transforms, and r rounds of pointer jumping compose 2^r of them
    rb.fk_rounds = 0;
    while ((1 << rb.fk_rounds) < maxd + 1) ++rb.fk_rounds;
  }
  for (int l = 0; l < d->n_links; ++l) {
    int f = d->link_frame[l];
    if (f < 0 || f >= d->n_frames) { delete h; return fail(nullptr, GTO_ERR_INVALID_ARG, "link_frame out of range"); }
    rb.link_frame[l] = f;
    rb.link_anc[l] = rb.frame_anc[f];
    double R[9];
    rpy2r(d->visual_rpy + 3 * l, R);
    rt2aff(R, d->visual_xyz + 3 * l, rb.vis_origin[l]);
  }
  for (int v = 0; v < 2; ++v) {
    const int NPv = v ? 16 : 8;
    for (int e = 0; e < NPv * NPv + NPv; ++e) {
      const int i = e < NPv * NPv ? e / NPv : e - NPv * NPv, j = e < NPv * NPv ? e % NPv : i;
      uint32_t m = 0u;
      if (i < d->n_opt && j < d->n_opt)
        for (int l = 0; l < d->n_links; ++l) m |= (((rb.link_anc[l] >> i) & (rb.link_anc[l] >> j) & 1u) << l);
      rb.entry_links[v][e] = m;
    }
  }
  for (int i = 0; i < d->n_frames; ++i) rb.link_of_frame[i] = -1, rb.xst_slot[i] = -1;
  for (int l = 0; l < d->n_links; ++l) {
    if (rb.link_of_frame[rb.link_frame[l]] >= 0) { delete h; return fail(nullptr, GTO_ERR_INVALID_ARG, "two collision links on one frame"); }
    rb.link_of_frame[rb.link_frame[l]] = l;
  }
  rb.n_xst = 0;
  rb.frame_free_mask = 0u;
  for (int i = 0; i < d->n_frames; ++i)
    if (rb.opt_of_frame[i] < 0) rb.frame_free_mask |= 1u << i;
  for (int i = 0; i < d->n_frames; ++i) {
    const int p = rb.parent[i];
    if (p >= 0 && p != i - 1 && rb.xst_slot[p] < 0) rb.xst_slot[p] = rb.n_xst++;
  }
  {  // the step kernel's walk over the tree (prebroad_tail): control words, and the widening of its culling radius
    for (int i = 0; i < d->n_frames; ++i) {
      const int p = rb.parent[i], l = rb.link_of_frame[i];
      const int src = p < 0 ? 0 : (p == i - 1 ? 1 : 2 + rb.xst_slot[p]);
      const bool moving_link = l >= 0 && (rb.frame_anc[rb.link_frame[l]] != 0u);
      rb.pb_ctl[i] = (src & 7) | (((rb.xst_slot[i] + 1) & 7) << 4) | ((moving_link ? 1 : 0) << 8);
      rb.pb_par[i] = -1;
      if (rb.joint_type[i] != GTO_JOINT_FIXED && rb.opt_of_frame[i] < 0) rb.pb_par[i] = rb.pb_npar, rb.pb_parf[rb.pb_npar++] = i;
    }
    double D = 0.0, maxp = 0.0;  // no frame origin or surface point is further than D from any other
    for (int i = 0; i < d->n_frames; ++i) {
      const double* ox = d->origin_xyz + 3 * i;
      D += std::sqrt(ox[0] * ox[0] + ox[1] * ox[1] + ox[2] * ox[2]);
      if (rb.joint_type[i] == GTO_JOINT_PRISMATIC) {
        const int j = rb.opt_of_frame[i];
        D += j >= 0 ? std::max(std::fabs(rb.lower[j]), std::fabs(rb.upper[j])) : 2.0;
      }
    }
    for (int l = 0; l < d->n_links; ++l) {
      const double* vx = d->visual_xyz + 3 * l;
      maxp = std::max(maxp, std::sqrt(vx[0] * vx[0] + vx[1] * vx[1] + vx[2] * vx[2]));
    }
    double maxr = 0.0;
    for (int i = 0; i < d->n_points; ++i) {
      const double* pp = d->points + 3 * i;
      maxr = std::max(maxr, std::sqrt(pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2]));
    }
    D += maxp + maxr;
    rb.pb_eps = 128.0 * d->n_frames * 5.9604644775390625e-08 * D;  // four times the first-order bound 32 F 2^-24 D (gto_kernels.h, PbLayout)
  }
  {  // operand tables of fk_mfma_tree: the full tree, and the compact tree of the obstacle kernel (gto_device.h)
    const int F = d->n_frames, L = d->n_links, n = d->n_opt;
    auto hom = [](const double* aff, int a, int c) { return a < 3 ? aff[4 * a + c] : (c == 3 ? 1.0 : 0.0); };
    // one table: nf frames (origin affine, unit axis, joint type, parent, optimised-joint slot), the links' frames and
    // visual origins, the optimised joints' frames
    auto build = [&](double* T0, int nf, const double (*org)[12], const double (*axu)[3], const int* jtype, const int* par,
                     const int* optj, const int* lframe, const double (*vorg)[12], int* opt_frame_out) {
      std::memset(T0, 0, sizeof rb.fk_tab);
      for (int i = 0; i < nf; ++i) {
        const double* u = axu[i];
        const double K[3][3] = {{0, -u[2], u[1]}, {u[2], 0, -u[0]}, {-u[1], u[0], 0}};
        double* tab = T0 + GTO_FK_STRIDE * i;
        for (int a = 0; a < 4; ++a)
          for (int c = 0; c < 4; ++c) {
            const int e = (4 * a + c) ^ (5 * (i & 3));  // bank placement of the frame's entries (gto_device.h, fkx)
            const double uu = (a < 3 && c < 3) ? u[a] * u[c] : 0.0;
            const double dl = (a == c && a < 3) ? 1.0 : 0.0, hh = (a == 3 && c == 3) ? 1.0 : 0.0;
            tab[e] = hom(org[i], c, a);  // transposed: a row-pattern read gives the B operand O^T (gto_device.h, fkx)
            tab[16 + e] = hh + uu;  // M = c0 + cos c1 + sin K
            tab[32 + e] = dl - uu;
            if (jtype[i] == GTO_JOINT_PRISMATIC) tab[48 + e] = (a < 3 && c == 3) ? u[a] : 0.0;
            else tab[48 + e] = (a < 3 && c < 3) ? K[a][c] : 0.0;
          }
        if (optj[i] >= 0) {
          const int j = optj[i];
          opt_frame_out[j] = i;
          double* tu = T0 + GTO_FK_STRIDE * nf + 16 * L;
          for (int a = 0; a < 3; ++a) tu[fkx(j, 4 * a)] = u[a];
          tu[fkx(j, 4 * 3 + 1)] = 1.0;
        }
      }
      for (int l = 0; l < L; ++l)
        for (int a = 0; a < 4; ++a)
          for (int c = 0; c < 4; ++c) T0[GTO_FK_STRIDE * nf + fkx(l, 4 * a + c)] = hom(vorg[l], a, c);
      double* tI = T0 + GTO_FK_STRIDE * nf + 16 * L + 16 * n;
      for (int l = 0; l < L; ++l) tI[l] = lframe[l];
      for (int j = 0; j < n; ++j) {
        tI[L + j] = opt_frame_out[j];
        tI[L + n + j] = jtype[opt_frame_out[j]] == GTO_JOINT_PRISMATIC ? 1.0 : 0.0;
      }
      for (int i = 0; i < nf; ++i) tI[L + 2 * n + i] = par[i];
    };
    build(rb.fk_tab, F, rb.origin, rb.axis_unit, rb.joint_type, rb.parent, rb.opt_of_frame, rb.link_frame, rb.vis_origin, rb.opt_frame);
    // the compact tree
    static_assert(sizeof rb.fk_tab == sizeof rb.fk_tab_c, "one layout for both tables");
    int cf[GTO_MAX_FRAMES], nc = 0;
    double corg[GTO_MAX_FRAMES][12], caxu[GTO_MAX_FRAMES][3], cvis[GTO_MAX_LINKS][12];
    int cpar[GTO_MAX_FRAMES], coptj[GTO_MAX_FRAMES], clf[GTO_MAX_LINKS], copt_frame[GTO_MAX_OPT];
    for (int f = 0; f < F; ++f) cf[f] = rb.joint_type[f] != GTO_JOINT_FIXED ? nc++ : -1;
    for (int f = 0; f < F; ++f) {
      if (cf[f] < 0) continue;
      const int k = cf[f];
      double A[12];
      std::memcpy(A, rb.origin[f], sizeof A);
      int p = rb.parent[f];
      for (; p >= 0 && rb.joint_type[p] == GTO_JOINT_FIXED; p = rb.parent[p]) aff_mul(rb.origin[p], A, A);
      std::memcpy(corg[k], A, sizeof A);
      std::memcpy(caxu[k], rb.axis_unit[f], sizeof caxu[k]);
      cpar[k] = p >= 0 ? cf[p] : -1;
      coptj[k] = rb.opt_of_frame[f];
      rb.cf_orig[k] = f;
      rb.cf_type[k] = rb.joint_type[f];
    }
    bool need_world = false;
    for (int l = 0; l < L; ++l) {
      double V[12];
      std::memcpy(V, rb.vis_origin[l], sizeof V);
      int g = rb.link_frame[l];
      for (; g >= 0 && rb.joint_type[g] == GTO_JOINT_FIXED; g = rb.parent[g]) aff_mul(rb.origin[g], V, V);
      std::memcpy(cvis[l], V, sizeof V);
      clf[l] = g >= 0 ? cf[g] : -2;
      need_world = need_world || g < 0;
    }
    if (need_world || nc == 0) {  // links that hang on fixed frames only: a frame that is the world
      const int k = nc++;
      const double I12[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      std::memcpy(corg[k], I12, sizeof I12);
      caxu[k][0] = caxu[k][1] = caxu[k][2] = 0.0;
      cpar[k] = -1, coptj[k] = -1;
      rb.cf_orig[k] = -1;
      rb.cf_type[k] = GTO_JOINT_FIXED;
      for (int l = 0; l < L; ++l)
        if (clf[l] == -2) clf[l] = k;
    }
    build(rb.fk_tab_c, nc, corg, caxu, rb.cf_type, cpar, coptj, clf, cvis, copt_frame);
    rb.n_cframes = nc;
    int depth[GTO_MAX_FRAMES], maxd = 1;
    for (int i = 0; i < nc; ++i) {
      depth[i] = cpar[i] < 0 ? 0 : depth[cpar[i]] + 1;
      maxd = std::max(maxd, depth[i]);
    }
    rb.fk_rounds_c = 0;
    while ((1 << rb.fk_rounds_c) < maxd + 1) ++rb.fk_rounds_c;
  }
  // moments of the gripper point cloud
  rb.grip_count = (double)d->n_gripper_points;
  for (int k = 0; k < d->n_gripper_points; ++k) {
    const double* p = d->gripper_points + 3 * k;
    for (int r = 0; r < 3; ++r) {
      rb.grip_mu[r] += p[r];
      for (int c = 0; c < 3; ++c) rb.grip_M[3 * r + c] += p[r] * p[c];
    }
  }
  // reach[j]: conservative bound on the displacement of any surface point per unit motion of joint j:
  // sum of the origin offsets along the chain below the joint (+ travel of prismatic joints below it)
  // + visual-origin offset + farthest surface point of the link, maximised over the links it moves.
  {
    std::vector<double> maxpt(d->n_links, 0.0);
    for (int i = 0; i < d->n_points; ++i) {
      const double* p = d->points + 3 * i;
      maxpt[d->point_link[i]] = std::max(maxpt[d->point_link[i]], std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
    }
    for (int j = 0; j < d->n_opt; ++j) rb.reach[j] = 0.0;
    for (int l = 0; l < GTO_MAX_LINKS; ++l)
      for (int j = 0; j < GTO_MAX_OPT; ++j) rb.reach_link[l][j] = 0.0;
    bool ok = true;
    for (int l = 0; l < d->n_links; ++l) {
      const double* vx = d->visual_xyz + 3 * l;
      double below = std::sqrt(vx[0] * vx[0] + vx[1] * vx[1] + vx[2] * vx[2]) + maxpt[l];  // offsets below the current frame
      for (int f = d->link_frame[l]; f >= 0; f = d->parent[f]) {
        const int j = rb.opt_of_frame[f];
        if (j >= 0) rb.reach[j] = std::max(rb.reach[j], rb.joint_type[f] == GTO_JOINT_PRISMATIC ? 1.0 : below);
        if (j >= 0) rb.reach_link[l][j] = rb.joint_type[f] == GTO_JOINT_PRISMATIC ? 1.0 : below;
        if (rb.joint_type[f] == GTO_JOINT_PRISMATIC) {
          // travel of this joint moves everything below it further from the joints above
          double travel = 1e9;
          if (j >= 0) travel = std::max(std::fabs(rb.lower[j]), std::fabs(rb.upper[j]));
          else travel = 2.0;  // parameter prismatic joints (torso lift, fingers): generous constant
          if (!(travel < 1e3)) ok = false;
          below += travel;
        }
        const double* ox = d->origin_xyz + 3 * f;
        below += std::sqrt(ox[0] * ox[0] + ox[1] * ox[1] + ox[2] * ox[2]);
      }
    }
    if (!ok) for (int j = 0; j < d->n_opt; ++j) rb.reach[j] = -1.0;
    if (!ok)
      for (int l = 0; l < GTO_MAX_LINKS; ++l)
        for (int j = 0; j < GTO_MAX_OPT; ++j) rb.reach_link[l][j] = -1.0;
  }
  // points sorted by link (stable), chunk table of <= 64 link-uniform points
  const int P = d->n_points;
  std::vector<int32_t> perm(P);
  for (int i = 0; i < P; ++i) {
    perm[i] = i;
    if (d->point_link[i] < 0 || d->point_link[i] >= d->n_links) { delete h; return fail(nullptr, GTO_ERR_INVALID_ARG, "point_link out of range"); }
  }
  // sort by link, and inside a link along a Morton (Z-order) curve of the link-local coordinates, so
  // that the 64 lanes of a chunk (and consecutive chunks) gather from neighbouring voxels/cache lines
  std::vector<uint32_t> morton(P, 0);
  {
    std::vector<double> lo(3 * d->n_links, 1e300), hi(3 * d->n_links, -1e300);
    for (int i = 0; i < P; ++i)
      for (int a = 0; a < 3; ++a) {
        lo[3 * d->point_link[i] + a] = std::min(lo[3 * d->point_link[i] + a], d->points[3 * i + a]);
        hi[3 * d->point_link[i] + a] = std::max(hi[3 * d->point_link[i] + a], d->points[3 * i + a]);
      }
    auto spread = [](uint32_t v) {  // 10 bits -> every third bit
      v &= 0x3ff;
      v = (v | (v << 16)) & 0x30000ff;
      v = (v | (v << 8)) & 0x300f00f;
      v = (v | (v << 4)) & 0x30c30c3;
      v = (v | (v << 2)) & 0x9249249;
      return v;
    };
    for (int i = 0; i < P; ++i) {
      const int l = d->point_link[i];
      double ext = 1e-12;
      for (int a = 0; a < 3; ++a) ext = std::max(ext, hi[3 * l + a] - lo[3 * l + a]);
      uint32_t code = 0;
      for (int a = 0; a < 3; ++a) {
        double u = (d->points[3 * i + a] - lo[3 * l + a]) / ext;  // isotropic cells
        uint32_t q = (uint32_t)std::min(1023.0, std::max(0.0, u * 1023.0));
        code |= spread(q) << a;
      }
      morton[i] = code;
    }
  }
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
    if (d->point_link[a] != d->point_link[b]) return d->point_link[a] < d->point_link[b];
    return morton[a] < morton[b];
  });
  std::vector<double> px(P), py(P), pz(P);
  std::vector<int32_t> plink(P);
  for (int i = 0; i < P; ++i) {
    px[i] = d->points[3 * perm[i]];
    py[i] = d->points[3 * perm[i] + 1];
    pz[i] = d->points[3 * perm[i] + 2];
    plink[i] = d->point_link[perm[i]];
  }
  std::vector<Chunk> chunks;
  for (int i = 0; i < P;) {
    int l = plink[i], j = i;
    while (j < P && plink[j] == l && j - i < GTO_WAVE) ++j;
    // bounding sphere: centre = mean of the chunk's points, radius = farthest point (+ a hair)
    double cx = 0, cy = 0, cz = 0;
    for (int k = i; k < j; ++k) { cx += px[k]; cy += py[k]; cz += pz[k]; }
    cx /= (j - i); cy /= (j - i); cz /= (j - i);
    double r2 = 0;
    for (int k = i; k < j; ++k) {
      double dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
      r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
    }
    chunks.push_back(Chunk{l, i, j - i, rb.link_anc[l] == 0u ? 1 : 0, cx, cy, cz, std::sqrt(r2) * (1.0 + 1e-9) + 1e-12});
    i = j;
  }
  rb.n_chunks = (int)chunks.size();
  std::vector<PbChunk> pbchunks;
  // The step kernel's spheres: runs of h->pb_merge consecutive chunks of a moving link under one sphere (the chunks follow a
  // Morton curve, so a run is a compact patch; centre = mean of its points, radius = the farthest of them).  Coarser than
  // the obstacle kernel's own test, which stays per chunk and exact: a sphere more lists a group more, never one less.
  // Centre in the coordinates of the link's FRAME (visual origin applied here, in double: the tail's walk stops at the frames).
  for (size_t ci = 0; ci < chunks.size();) {
    const Chunk& c = chunks[ci];
    if (c.pad) { ++ci; continue; }
    size_t cj = ci;
    int i0 = c.start, i1 = c.start;
    while (cj < chunks.size() && cj - ci < (size_t)h->pb_merge && !chunks[cj].pad && chunks[cj].link == c.link) i1 = chunks[cj].start + chunks[cj].count, ++cj;
    double cx = 0, cy = 0, cz = 0, r2 = 0;
    for (int k = i0; k < i1; ++k) cx += px[k], cy += py[k], cz += pz[k];
    cx /= (i1 - i0), cy /= (i1 - i0), cz /= (i1 - i0);
    for (int k = i0; k < i1; ++k) r2 = std::max(r2, (px[k] - cx) * (px[k] - cx) + (py[k] - cy) * (py[k] - cy) + (pz[k] - cz) * (pz[k] - cz));
    const double* V = rb.vis_origin[c.link];
    pbchunks.push_back(PbChunk{V[0] * cx + V[1] * cy + V[2] * cz + V[3], V[4] * cx + V[5] * cy + V[6] * cz + V[7], V[8] * cx + V[9] * cy + V[10] * cz + V[11],
                               std::sqrt(r2) * (1.0 + 1e-9) + 1e-12, rb.link_frame[c.link], 0});
    ci = cj;
  }
  h->pb_C = (int)pbchunks.size();
  if (pbchunks.empty()) pbchunks.push_back(PbChunk{0, 0, 0, 0, 0, 0});  // (a robot none of whose links moves: the table is never read)
  if (rb.n_chunks > GTO_MAX_ACTIVE) { delete h; return fail(nullptr, GTO_ERR_UNSUPPORTED, "too many surface points (max 16384)"); }

  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail(nullptr, GTO_ERR_HIP, "hipStreamCreate failed"); }
  auto up = [&](void** dst, const void* src, size_t bytes) -> bool {
    if (hipMalloc(dst, bytes) != hipSuccess) return false;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = up((void**)&h->d_rb, &rb, sizeof rb) && up((void**)&h->d_px, px.data(), P * sizeof(double)) &&
            up((void**)&h->d_py, py.data(), P * sizeof(double)) && up((void**)&h->d_pz, pz.data(), P * sizeof(double)) &&
            up((void**)&h->d_plink, plink.data(), P * sizeof(int32_t)) && up((void**)&h->d_perm, perm.data(), P * sizeof(int32_t)) &&
            up((void**)&h->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk)) &&
            up((void**)&h->d_pbchunks, pbchunks.data(), std::max<size_t>(1, pbchunks.size()) * sizeof(PbChunk));
  if (!ok) { gto_destroy(h); return fail(nullptr, GTO_ERR_ALLOC, "device allocation failed in gto_create"); }
  h->np = rb.n_opt <= GTO_NB ? GTO_NB : 16;
  h->lm_lds = h->np == GTO_NB ? lm_lds_bytes(opts->T, 1) : lm_wide_lds_bytes(opts->T, 16);
  if (h->lm_lds > 160 * 1024) { gto_destroy(h); return fail(nullptr, GTO_ERR_UNSUPPORTED, "T too large for the step kernel's LDS"); }
  // candidates per step the eight-wave step kernel's LDS has room for at this T
  h->spec_kmax = 1;
  if (h->np == GTO_NB)
    while (h->spec_kmax < GTO_KSPEC && lm_lds_bytes(opts->T, h->spec_kmax + 1) <= 160 * 1024 - 256) ++h->spec_kmax;
  {
    const int w = h->np == GTO_NB ? 0 : 1;
    // waypoints per obstacle workgroup of the wide robots (see below), fewer if the robot's tables would not fit the CU's LDS
    int tg_w = getenv("GTO_OBS_TG_WIDE") ? std::max(1, std::min(GTO_MAX_TG, atoi(getenv("GTO_OBS_TG_WIDE")))) : 5;
    if (w && getenv("GTO_OBS_TG")) tg_w = std::max(tg_w, h->obs_tg);  // (GTO_OBS_TG, when given, is the group size of every robot)
    while (w && tg_w > 1 && (size_t)ObsLds(tg_w, rb.n_frames, rb.n_links, tg_w * rb.n_chunks, h->np).total_doubles * sizeof(double) > 150 * 1024) --tg_w;
    const int tg_lds = w ? tg_w : GTO_MAX_TG;
    const ObsLds lay(tg_lds, rb.n_frames, rb.n_links, tg_lds * rb.n_chunks, h->np);
    const size_t lds = std::min<size_t>((size_t)lay.total_doubles * sizeof(double), 160 * 1024);
    if ((size_t)lay.total_doubles * sizeof(double) > 150 * 1024) { gto_destroy(h); return fail(nullptr, GTO_ERR_UNSUPPORTED, "robot too large for the obstacle kernel's LDS"); }
    hipError_t e1 = raise_dynamic_lds(w ? (const void*)k_obstacle_gram<16> : (const void*)k_obstacle_gram<GTO_NB>, lds);
    if (!w && e1 == hipSuccess) e1 = raise_dynamic_lds((const void*)k_obstacle_gram<GTO_NB, GTO_OBS_DEEP_PD>, lds);
    if (!w && e1 == hipSuccess) e1 = raise_dynamic_lds((const void*)k_obstacle_gram<GTO_NB, GTO_OBS_MAIN_PD, true>, lds);
    if (e1 == hipSuccess) e1 = raise_dynamic_lds(w ? (const void*)k_obstacle_gram<16, GTO_OBS_MAIN_PD, false, true> : (const void*)k_obstacle_gram<GTO_NB, GTO_OBS_MAIN_PD, false, true>, lds);
    if (!w && e1 == hipSuccess) e1 = raise_dynamic_lds((const void*)k_obstacle_gram<GTO_NB, GTO_OBS_DEEP_PD, false, true>, lds);
    hipError_t e2 = w ? raise_dynamic_lds((const void*)k_lm_step_wide<16>, h->lm_lds) : raise_dynamic_lds((const void*)k_lm_step<4, 1>, h->lm_lds);
    if (!w && e2 == hipSuccess) e2 = raise_dynamic_lds((const void*)k_lm_step<8, GTO_KSPEC>, lm_lds_bytes(opts->T, h->spec_kmax));
    if (e1 != hipSuccess || e2 != hipSuccess) {
      gto_destroy(h);
      return fail(nullptr, GTO_ERR_HIP, "hipFuncSetAttribute failed");
    }
    if (w) {  // wider blocks (GTO_OBS_TG_WIDE): five waypoints per workgroup.  The fixed part of a 16-wide workgroup (tables, matrix-core
      // prefix, projection onto sixteen screws) is larger than an 8-wide one's, and since round 6 the epilogue projects one
      // waypoint at a time (ObsLds: its scratch no longer grows with the group): configs[4] with 2 / 3 / 4 / 5 / 6 / 8 waypoints
      // per workgroup 29.0 / 33.1 / 35.4 / 36.6 / 34.8 / 28.7 k trajectories/s (until round 6: two, 28.2 k)
      h->obs_tg = getenv("GTO_OBS_TG") ? std::min(h->obs_tg, tg_w) : tg_w;
      h->obs_tg_few = std::min(h->obs_tg_few, h->obs_tg), h->obs_tg_few_tail = std::min(h->obs_tg_few_tail, h->obs_tg);
    }
  }
  *out = h;
  return GTO_OK;
}

void gto_destroy(gto_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& s : h->scenes) {
    if (s.valid == 1) {
      if (s.c_obs != s.c_all) (void)hipFree((void*)s.c_obs);
      (void)hipFree((void*)s.c_all);
      if (s.r_obs != s.r_all) (void)hipFree((void*)s.r_obs);
      (void)hipFree((void*)s.r_all);
      if (s.d_obs != s.d_all) (void)hipFree((void*)s.d_obs);
      (void)hipFree((void*)s.d_all);
    }
  }
  (void)hipFree(h->d_scenes);
  (void)hipFree(h->d_rb);
  (void)hipFree(h->d_px);
  (void)hipFree(h->d_py);
  (void)hipFree(h->d_pz);
  (void)hipFree(h->d_plink);
  (void)hipFree(h->d_perm);
  (void)hipFree(h->d_chunks);
  (void)hipFree(h->d_pbchunks);
  DevBuf* bufs[] = {&h->zws, &h->counters, &h->state, &h->Qcur, &h->Qtry, &h->vis, &h->screw, &h->blocks, &h->goalblk, &h->ssfixed, &h->ndone, &h->qf, &h->livebuf, &h->qfs, &h->wrecbuf, &h->itembuf};
  if (h->h_ndone) (void)hipHostFree(h->h_ndone);
  if (h->h_progress) (void)hipHostFree(h->h_progress);
  for (DevBuf* b : bufs) (void)hipFree(b->p);
  for (auto& sp_ : h->spare) (void)hipFree(sp_.first);
  for (auto& b : h->in) (void)hipFree(b.p);
  for (auto& b : h->out) (void)hipFree(b.p);
  for (auto& b : h->pin_in) if (b.p) (void)hipHostFree(b.p);
  for (auto& b : h->pin_out) if (b.p) (void)hipHostFree(b.p);
  for (auto e : h->ev) (void)hipEventDestroy(e);
  if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
  for (int l = 0; l < GTO_MAX_LANES; ++l) {
    if (h->lane_stream[l]) (void)hipStreamDestroy(h->lane_stream[l]);
    if (h->lane_event[l]) (void)hipEventDestroy(h->lane_event[l]);
  }
  delete h;
}

int gto_set_opts(gto_handle* h, const gto_solver_opts* o) {
  if (!h || !o) return GTO_ERR_INVALID_ARG;
  std::string why;
  if (!validate_opts(o, why)) return fail(h, GTO_ERR_INVALID_ARG, why);
  if (o->T != h->opts.T) return fail(h, GTO_ERR_INVALID_ARG, "T cannot change after gto_create");
  h->opts = *o;
  return GTO_OK;
}

static int sync_scene_table(gto_handle* h) {
  size_t n = h->scenes.size();
  if (n > h->d_scenes_cap) {
    if (h->d_scenes) HIPCHK(h, hipFree(h->d_scenes));
    h->d_scenes = nullptr;
    size_t cap = std::max<size_t>(16, n * 2);
    HIPCHK(h, hipMalloc((void**)&h->d_scenes, cap * sizeof(SceneDev)));
    h->d_scenes_cap = cap;
  }
  HIPCHK(h, hipMemcpy(h->d_scenes, h->scenes.data(), n * sizeof(SceneDev), hipMemcpyHostToDevice));
  return GTO_OK;
}

// valid: 0 empty, 1 owned by this handle, 2 borrowed from another handle (gto_share_scene)
static int free_scene(gto_handle* h, SceneDev& s) {
  if (s.valid == 1) {
    if (s.c_obs != s.c_all) HIPCHK(h, hipFree((void*)s.c_obs));
    HIPCHK(h, hipFree((void*)s.c_all));
    if (s.r_obs != s.r_all) HIPCHK(h, hipFree((void*)s.r_obs));
    HIPCHK(h, hipFree((void*)s.r_all));
    if (s.d_obs != s.d_all) HIPCHK(h, hipFree((void*)s.d_obs));
    HIPCHK(h, hipFree((void*)s.d_all));
  }
  s.valid = 0;
  return GTO_OK;
}

int gto_share_scene(gto_handle* dst, int32_t dst_id, gto_handle* src, int32_t src_id) {
  return gto_share_scene_halves(dst, dst_id, src, src_id, 0, 1);
}

int gto_share_scene_halves(gto_handle* dst, int32_t dst_id, gto_handle* src, int32_t src_id, int32_t all_from, int32_t obs_from) {
  if (!dst || !src) return GTO_ERR_INVALID_ARG;
  if ((all_from != 0 && all_from != 1) || (obs_from != 0 && obs_from != 1)) return fail(dst, GTO_ERR_INVALID_ARG, "gto_share_scene_halves: a half is 0 (c_all) or 1 (c_obs)");
  if (dst_id < 0 || dst_id >= 65536) return fail(dst, GTO_ERR_INVALID_ARG, "scene_id out of range [0,65536)");
  if (src_id < 0 || (size_t)src_id >= src->scenes.size() || !src->scenes[src_id].valid)
    return fail(dst, GTO_ERR_NO_SCENE, "gto_share_scene: the source scene was never set");
  if (dst->device != src->device) return fail(dst, GTO_ERR_INVALID_ARG, "gto_share_scene: handles live on different devices");
  HIPCHK(dst, hipSetDevice(dst->device));
  HIPCHK(dst, hipStreamSynchronize(dst->stream));
  if ((size_t)dst_id >= dst->scenes.size()) {
    SceneDev z;
    memset(&z, 0, sizeof z);
    dst->scenes.resize(dst_id + 1, z);
  }
  int rcf = free_scene(dst, dst->scenes[dst_id]);
  if (rcf) return rcf;
  const SceneDev& ss = src->scenes[src_id];
  SceneDev& ds = dst->scenes[dst_id];
  ds = ss;
  ds.c_all = all_from ? ss.c_obs : ss.c_all, ds.r_all = all_from ? ss.r_obs : ss.r_all, ds.d_all = all_from ? ss.d_obs : ss.d_all;
  ds.c_obs = obs_from ? ss.c_obs : ss.c_all, ds.r_obs = obs_from ? ss.r_obs : ss.r_all, ds.d_obs = obs_from ? ss.d_obs : ss.d_all;
  ds.valid = 2;
  return sync_scene_table(dst);
}

static int set_scene_impl(gto_handle* h, int32_t id, const float* c_all, const float* c_obs, const int32_t shape[3],
                          const double origin[3], double res, bool values_only, hipMemcpyKind kind = hipMemcpyHostToDevice) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (!c_all || !shape || !origin) return fail(h, GTO_ERR_INVALID_ARG, "null argument");
  if (id < 0 || id >= 65536) return fail(h, GTO_ERR_INVALID_ARG, "scene_id out of range [0,65536)");
  if (shape[0] < 1 || shape[1] < 1 || shape[2] < 1 || !(res > 0)) return fail(h, GTO_ERR_INVALID_ARG, "bad field geometry");
  // (the gather loop forms the voxel offset with 24-bit multiply-adds: iz + nz (iy + ny ix))
  if ((long long)shape[0] * shape[1] > (1ll << 24) || shape[2] >= (1 << 24) || (long long)shape[0] * shape[1] * shape[2] >= (1ll << 32))
    return fail(h, GTO_ERR_UNSUPPORTED, "field too large: nx ny <= 2^24, nz < 2^24 and fewer than 2^32 voxels");
  const size_t nvox = (size_t)shape[0] * shape[1] * shape[2];
  if (nvox >= ((size_t)1 << 31)) return fail(h, GTO_ERR_UNSUPPORTED, "field larger than 2^31 voxels");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  // Build the new scene completely first (fields, voxel records, distance fields), then swap it in and free the old one:
  // a failure half-way leaves the table and the previous scene of this id untouched and frees what was allocated.
  std::vector<void*> owned;
  auto fail_free = [&](hipError_t e, const char* what) {
    for (void* p : owned) (void)hipFree(p);
    h->err = std::string(what) + ": " + hipGetErrorString(e);
    return GTO_ERR_HIP;
  };
#define SCN(call)                                      \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return fail_free(e_, #call); \
  } while (0)
  auto dalloc = [&](void** p, size_t bytes) {
    for (size_t k = 0; k < h->spare.size(); ++k)
      if (h->spare[k].second == bytes) {  // a buffer of the scene replaced last time
        *p = h->spare[k].first;
        h->spare.erase(h->spare.begin() + k);
        owned.push_back(*p);
        return hipSuccess;
      }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) owned.push_back(*p);
    return e;
  };
  float *da = nullptr, *dob = nullptr;
  SCN(dalloc((void**)&da, nvox * sizeof(float)));
  SCN(hipMemcpy(da, c_all, nvox * sizeof(float), kind));
  if (c_obs && c_obs != c_all) {
    SCN(dalloc((void**)&dob, nvox * sizeof(float)));
    SCN(hipMemcpy(dob, c_obs, nvox * sizeof(float), kind));
  } else {
    dob = da;
  }
  // gather-friendly voxel records (one-time, outside the timed solve); a values-only scene (gto_set_scene_values: seed
  // scoring and point look-ups) stops at the two float fields
  VoxelRec *ra = nullptr, *rob = nullptr;
  uint8_t *dista = nullptr, *distb = nullptr, *scratch = nullptr;
  if (!values_only) {
  SCN(dalloc((void**)&ra, nvox * sizeof(VoxelRec)));
  const unsigned nblk = (unsigned)((nvox + 255) / 256);
  hipLaunchKernelGGL(k_build_records, dim3(nblk), dim3(256), 0, h->stream, da, ra, shape[0], shape[1], shape[2]);
  if (dob != da) {
    SCN(dalloc((void**)&rob, nvox * sizeof(VoxelRec)));
    hipLaunchKernelGGL(k_build_records, dim3(nblk), dim3(256), 0, h->stream, dob, rob, shape[0], shape[1], shape[2]);
  } else {
    rob = ra;
  }
  // broad-phase distance fields: ping-pong relaxation, the scratch half is freed again
  SCN(dalloc((void**)&scratch, nvox));
  for (int which = 0; which < (rob != ra ? 2 : 1); ++which) {
    uint8_t* d0 = nullptr;
    SCN(dalloc((void**)&d0, nvox));
    if (h->dist_relax) {  // GTO_DIST_RELAX=1: the reference construction, GTO_DIST_CAP sweeps of 3x3x3 min-plus-one (A/B and tests)
      uint8_t* d1 = scratch;
      hipLaunchKernelGGL(k_dist_init, dim3(nblk), dim3(256), 0, h->stream, which ? rob : ra, d0, (long)nvox);
      for (int it = 0; it < GTO_DIST_CAP; ++it) {
        hipLaunchKernelGGL(k_dist_relax, dim3(nblk), dim3(256), 0, h->stream, d0, d1, shape[0], shape[1], shape[2]);
        std::swap(d0, d1);
      }
      static_assert(GTO_DIST_CAP % 2 == 0, "ping-pong parity: the result is back in the buffer allocated for it");
    } else {  // separable: one exact pass per axis, scratch -> d0 -> scratch -> d0
      hipLaunchKernelGGL(k_dist_init, dim3(nblk), dim3(256), 0, h->stream, which ? rob : ra, scratch, (long)nvox);
      hipLaunchKernelGGL(k_dist_axis, dim3(nblk), dim3(256), 0, h->stream, scratch, d0, shape[0], shape[1], shape[2], 2);
      hipLaunchKernelGGL(k_dist_axis, dim3(nblk), dim3(256), 0, h->stream, d0, scratch, shape[0], shape[1], shape[2], 1);
      hipLaunchKernelGGL(k_dist_axis, dim3(nblk), dim3(256), 0, h->stream, scratch, d0, shape[0], shape[1], shape[2], 0);
    }
    (which ? distb : dista) = d0;
  }
  if (rob == ra) distb = dista;
  }
  SCN(hipStreamSynchronize(h->stream));
  SCN(hipGetLastError());
#undef SCN
  if ((size_t)id >= h->scenes.size()) {
    SceneDev z;
    memset(&z, 0, sizeof z);
    h->scenes.resize(id + 1, z);
  }
  SceneDev& s = h->scenes[id];
  // what is left of the previous spares did not fit this scene: free it; the replaced scene's buffers become the spares
  for (auto& sp_ : h->spare) (void)hipFree(sp_.first);
  h->spare.clear();
  if (scratch) h->spare.emplace_back((void*)scratch, nvox);
  if (s.valid == 1) {
    const size_t ov = (size_t)s.nx * s.ny * s.nz;
    auto keep = [&](const void* p_, size_t bytes) {
      if (p_) h->spare.emplace_back(const_cast<void*>(p_), bytes);
    };
    if (s.c_obs != s.c_all) keep(s.c_obs, ov * sizeof(float));
    keep(s.c_all, ov * sizeof(float));
    if (s.r_obs != s.r_all) keep(s.r_obs, ov * sizeof(VoxelRec));
    keep(s.r_all, ov * sizeof(VoxelRec));
    if (s.d_obs != s.d_all) keep(s.d_obs, ov);
    keep(s.d_all, ov);
    s.valid = 0;
  }
  int rcf = free_scene(h, s);
  if (rcf) return rcf;
  s.c_all = da;
  s.c_obs = dob;
  s.r_all = ra;
  s.r_obs = rob;
  s.d_all = dista;
  s.d_obs = distb;
  s.nx = shape[0];
  s.ny = shape[1];
  s.nz = shape[2];
  s.ox = origin[0];
  s.oy = origin[1];
  s.oz = origin[2];
  s.res = res;
  s.rinv = 1.0 / res;
  s.inv2r = 1.0 / (2.0 * res);
  s.valid = 1;
  return sync_scene_table(h);
}

int gto_set_scene(gto_handle* h, int32_t id, const float* c_all, const float* c_obs, const int32_t shape[3],
                  const double origin[3], double res) {
  return set_scene_impl(h, id, c_all, c_obs, shape, origin, res, false);
}

int gto_set_scene_values(gto_handle* h, int32_t id, const float* c_all, const float* c_obs, const int32_t shape[3],
                         const double origin[3], double res) {
  return set_scene_impl(h, id, c_all, c_obs, shape, origin, res, true);
}

int gto_drop_scene(gto_handle* h, int32_t id) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (id < 0 || (size_t)id >= h->scenes.size() || !h->scenes[id].valid) return fail(h, GTO_ERR_NO_SCENE, "unknown scene");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  SceneDev& s = h->scenes[id];
  int rcf = free_scene(h, s);
  if (rcf) return rcf;
  memset(&s, 0, sizeof s);
  return sync_scene_table(h);
}

int gto_set_mode(gto_handle* h, int32_t mode) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (mode != GTO_MODE_ROUNDS && mode != GTO_MODE_SINGLE_LAUNCH) return fail(h, GTO_ERR_INVALID_ARG, "unknown solver mode");
  // (the single-launch kernel of rounds 1-3, one workgroup running an instance's whole solve, was 2.4-2.9x slower than the
  // rounds and had been a test-only second implementation since round 3: removed in round 4; the mode number stays reserved)
  if (mode == GTO_MODE_SINGLE_LAUNCH) return fail(h, GTO_ERR_UNSUPPORTED, "GTO_MODE_SINGLE_LAUNCH was removed: the rounds mode is the solver");
  h->mode = mode;
  return GTO_OK;
}

int gto_set_lanes(gto_handle* h, int32_t max_lanes, int32_t min_per_lane, int32_t adopt_below) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (max_lanes < 1 || max_lanes > GTO_MAX_LANES || min_per_lane < 1 || adopt_below < 0)
    return fail(h, GTO_ERR_INVALID_ARG, "gto_set_lanes: 1 <= max_lanes <= 8, min_per_lane >= 1, adopt_below >= 0");
  h->lanes_max = max_lanes, h->lane_min = min_per_lane, h->adopt_below = adopt_below;
  return GTO_OK;
}

int gto_set_lane_streams(gto_handle* h, int32_t n, void* const* streams) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (n < 0 || n > GTO_MAX_LANES || (n > 0 && !streams)) return fail(h, GTO_ERR_INVALID_ARG, "gto_set_lane_streams: 0 <= n <= 8 streams");
  for (int i = 0; i < n; ++i)
    if (!streams[i]) return fail(h, GTO_ERR_INVALID_ARG, "gto_set_lane_streams: null stream");
  for (int i = 0; i < n; ++i) h->user_lane_stream[i] = (hipStream_t)streams[i];
  h->n_user_lane_streams = n;
  return GTO_OK;
}

int gto_set_stream(gto_handle* h, void* stream) {
  if (!h) return GTO_ERR_INVALID_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  if (h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->stream && h->own_stream) HIPCHK(h, hipStreamDestroy(h->stream));
  h->stream = nullptr;
  if (stream) {
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
  } else {
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return GTO_OK;
}

int gto_last_kernel_work(gto_handle* h, uint64_t* points_gathered, uint64_t* chunk_tests) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (points_gathered) *points_gathered = h->last_counters[0];
  if (chunk_tests) *chunk_tests = h->last_counters[1];
  return GTO_OK;
}

int gto_last_kernel_profile(gto_handle* h, int32_t variant, double* total_ms, int32_t* launches, uint64_t* workgroups, uint64_t* points_gathered) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (variant < 0 || variant >= GTO_PROF_VARIANTS) return fail(h, GTO_ERR_INVALID_ARG, "unknown kernel variant");
  if (total_ms) *total_ms = h->prof_ms[variant];
  if (launches) *launches = (int32_t)h->prof_launches[variant];
  if (workgroups) *workgroups = (uint64_t)h->prof_wgs[variant];
  if (points_gathered) *points_gathered = h->prof_points[variant];
  return GTO_OK;
}

int gto_set_profiling(gto_handle* h, int32_t enabled) {
  if (!h) return GTO_ERR_INVALID_ARG;
  h->profiling = enabled != 0;
  return GTO_OK;
}

int gto_last_kernel_time(gto_handle* h, double* total_ms, int32_t* launches) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (total_ms) *total_ms = h->last_ms;
  if (launches) *launches = h->last_launches;
  return GTO_OK;
}

// -------------------------------------------------------------------------------------------------
static SolveParams make_params(const gto_handle* h, int n_max, bool use_standoff) {
  const gto_solver_opts& o = h->opts;
  SolveParams sp;
  sp.T = o.T;
  sp.ts = o.T + o.standoff_offset;
  sp.use_standoff = use_standoff ? 1 : 0;
  sp.n_max = n_max;
  sp.max_iter = o.max_iter;
  sp.grad_mode = o.grad_mode;
  sp.dt = o.Tmax / (double)(o.T - 1);  // gto/gto_planner.py:27-28
  sp.alpha = o.w_vel / (sp.dt * sp.dt);
  sp.w_obstacle = o.w_obstacle;
  sp.w_vel = o.w_vel;
  sp.tol_step = o.tol_step;
  sp.tol_rel_f = o.tol_rel_f;
  sp.lambda0 = o.lambda0;
  sp.dbg_cut = h->dbg_cut;
  sp.interleave = h->obs_interleave == 1;
  sp.static_pos = 0;
  sp.pb_next = 0, sp.pb_tg = 1, sp.pb_ng = 1, sp.pb_pw = 1, sp.pb_verify = 0;
  sp.pb_C = h->pb_C, sp.pb_tab0 = 0, sp.pb_mC = ObsGeom::magic(std::max(1, h->pb_C)), sp.pb_mF = ObsGeom::magic(h->rb.n_frames);
  sp.pb_eps = h->rb.pb_eps;
  sp.pb_npar = h->rb.pb_npar;
  sp.round = sp.parity = 0;
  sp.kcap = h->np == GTO_NB ? GTO_KSPEC : 1;  // candidate copies of the workspace (the wide step kernel generates one)
  sp.k_acc = sp.k_rej = sp.k_eval = 1;
  sp.spec_streak = h->spec_streak;
  return sp;
}

static int ensure_workspace(gto_handle* h, int B) {
  const RobotDev& rb = h->rb;
  const size_t T = h->opts.T, n = rb.n_opt;
  int rc;
  if ((rc = ensure(h, h->state, (size_t)B * sizeof(InstState)))) return rc;
  if ((rc = ensure(h, h->Qcur, (size_t)B * n * T * sizeof(double)))) return rc;
  const size_t kcap = h->np == GTO_NB ? GTO_KSPEC : 1;
  if ((rc = ensure(h, h->Qtry, kcap * B * n * T * sizeof(double)))) return rc;
  const size_t bstride = (size_t)h->np * h->np + h->np + 8;
  if ((rc = ensure(h, h->blocks, (kcap + 1) * B * T * bstride * sizeof(double)))) return rc;
  if ((rc = ensure(h, h->goalblk, (kcap + 1) * B * 2 * bstride * sizeof(double)))) return rc;
  if ((rc = ensure(h, h->ssfixed, (size_t)B * 4 * sizeof(double)))) return rc;
  if ((rc = ensure(h, h->ndone, 64 * GTO_MAX_LANES))) return rc;  // a finished-counter per lane, a cache line apart
  if ((rc = ensure(h, h->qf, (size_t)B * T * rb.n_frames * sizeof(double)))) return rc;
  if ((rc = ensure(h, h->wrecbuf, (size_t)(kcap + 1) * B * T * 8 * sizeof(double)))) return rc;
  if (!h->h_ndone) HIPCHK(h, hipHostMalloc((void**)&h->h_ndone, 64));
  if (!h->h_progress) {
    HIPCHK(h, hipHostMalloc((void**)&h->h_progress, 64 * GTO_MAX_LANES, hipHostMallocMapped));  // eight words per lane
    std::memset(h->h_progress, 0, 64 * GTO_MAX_LANES);  // every word the solve loop reads carries a call tag (never 0)
    HIPCHK(h, hipHostGetDevicePointer((void**)&h->d_progress, h->h_progress, 0));
  }
  return GTO_OK;
}

static BatchPtrs make_ptrs(gto_handle* h, const int32_t* scene_id, const double* qc, const double* goals,
                           const int32_t* n_goals, const double* standoff, const double* base_pos, const double* Q0) {
  BatchPtrs bp = {};
  bp.scene_id = scene_id;
  bp.qc = qc;
  bp.goals = goals;
  bp.n_goals = n_goals;
  bp.standoff = standoff;
  bp.base_pos = base_pos;
  bp.Q0 = Q0;
  bp.state = (InstState*)h->state.p;
  bp.Qcur = (double*)h->Qcur.p;
  bp.Qtry = (double*)h->Qtry.p;
  bp.blocks = (double*)h->blocks.p;
  bp.goalblk = (double*)h->goalblk.p;
  bp.ss_fixed = (double*)h->ssfixed.p;
  bp.n_done = (int32_t*)h->ndone.p;
  bp.qf = (double*)h->qf.p;
  bp.live = nullptr;  // the live lists only exist inside the solve loop
  bp.jobs = nullptr;
  bp.nlive = nullptr;
  bp.next = nullptr;
  bp.qfs = nullptr;
  bp.wrec = (double*)h->wrecbuf.p;
  bp.items = nullptr;
  bp.scenes = h->d_scenes;
  bp.cap = 0;
  bp.n_total = 0;
  bp.dbg = h->dbg;
  bp.work = nullptr;
  return bp;
}

static inline int obstacle_grid(int B, int nG) { return 8 * ((B + 7) / 8) * nG; }

// profiling (gto_set_profiling): a pair of HIP events on the launch stream around launch number h->last_launches
static int prof_begin(gto_handle* h, hipStream_t st, int variant, long long wgs) {
  if (h->prof_mu) h->prof_mu->lock();  // lanes of one call record into one list: begin .. end is one critical section
  const size_t need = (size_t)(h->last_launches + 1) * 2;
  hipError_t e = hipSuccess;
  while (h->ev.size() < need && e == hipSuccess) {
    hipEvent_t ev_;
    if ((e = hipEventCreate(&ev_)) == hipSuccess) h->ev.push_back(ev_);
  }
  if (e == hipSuccess) {
    if (h->ev_variant.size() <= (size_t)h->last_launches) h->ev_variant.resize(h->last_launches + 1), h->ev_wgs.resize(h->last_launches + 1);
    h->ev_variant[h->last_launches] = variant;
    h->ev_wgs[h->last_launches] = wgs;
    e = hipEventRecord(h->ev[2 * h->last_launches], st);
  }
  if (e != hipSuccess && h->prof_mu) h->prof_mu->unlock();
  HIPCHK(h, e);
  return GTO_OK;
}
static int prof_end(gto_handle* h, hipStream_t st) {
  const hipError_t e = hipEventRecord(h->ev[2 * h->last_launches + 1], st);
  h->last_launches++;
  if (h->prof_mu) h->prof_mu->unlock();
  HIPCHK(h, e);
  return GTO_OK;
}

static int launch_obstacle(gto_handle* h, hipStream_t st, const BatchPtrs& bp, const SolveParams& sp, int B, int t_begin,
                           int nT, int fixed_mode, bool timed, bool with_goal_terms = false, int n_jobs = 0, int tg = 0, bool deep = false,
                           bool itemized = false, int items_hint = 0) {
  // waypoints per workgroup: groups of h->obs_tg (the two pinned waypoints form one group)
  const int TG = fixed_mode ? 1 : std::max(1, std::min(tg > 0 ? tg : h->obs_tg, nT));  // the init pass has 4 virtual waypoints
  const ObsGeom geo(h->rb.n_cframes, h->rb.n_frames, h->rb.fk_rounds_c, h->rb.n_links, h->rb.n_opt, h->rb.n_chunks, TG, nT, h->np);
  const int nG = geo.nG;
  const int nb = n_jobs > 0 ? n_jobs : B;  // workgroups are laid out for the evaluation jobs there can be; B stays the batch (strides)
  int n_regular = obstacle_grid(nb, nG);
  // Itemized launches: laid out over the caller's estimate of the item list's length instead of its upper bound (nine
  // tenths of the workgroups of the upper bound find no item and leave; they cost the other lanes' launches dispatch
  // slots: +7 % trajectories/s without them); a crew of GTO_SWEEP_WGS workgroups behind the launch walks whatever the
  // estimate missed (the kernel's SWEEP variant), so the result does not depend on it.
  const bool sweep = itemized && items_hint > 0 && items_hint + GTO_SWEEP_WGS < n_regular && bp.live != nullptr && !fixed_mode && !deep && h->np == GTO_NB;  // (the wide robots' launches are never itemized)
  if (sweep) n_regular = 8 * ((items_hint + 7) / 8);
  const size_t lds = (size_t)geo.lay.total_doubles * sizeof(double);
  const dim3 grid(n_regular + (with_goal_terms ? 8 * ((nb + 31) / 32) : 0));  // goal-term jobs: four to a workgroup, in front, a multiple of eight workgroups
  const bool deep_v = h->np == GTO_NB && deep;
  if (timed) {
    int rc_ = prof_begin(h, st, deep_v ? GTO_PROF_OBSTACLE_FEW : GTO_PROF_OBSTACLE, (long long)grid.x);
    if (rc_) return rc_;
  }
  BatchPtrs bpl = bp;  // the work counters of this variant (64 cells each)
  if (bpl.work) bpl.work += 64 * (deep_v ? GTO_PROF_OBSTACLE_FEW : GTO_PROF_OBSTACLE);
  // this round's job list and its length (the kernel's first, preloaded, arguments); null outside the solve loop
  const bool listed = bp.live != nullptr && !fixed_mode;
  const int32_t* jobs_par = listed ? bp.jobs + (size_t)sp.parity * bp.cap * sp.kcap : nullptr;
  const int32_t* njobs_par = listed ? bp.nlive + GTO_NJOBS(sp.parity) : nullptr;
  // rounds behind a step kernel that ran the broad phase itself: the regular workgroups are laid out over its list of (job, group) pairs
  const size_t items_cap = (size_t)bp.cap * sp.kcap * (sp.T - 2) + GTO_ITEM_SLACK;
  const int2* items_par = listed && itemized ? bp.items + (size_t)sp.parity * items_cap : nullptr;
  const int32_t* nitems_par = listed && itemized ? bp.nlive + 8 + sp.parity : nullptr;
  const bool hot = !fixed_mode && sp.grad_mode == GTO_GRAD_CENTRAL_DIFF && h->hot_variants;
  if (hot && h->np == GTO_NB && deep)
    hipLaunchKernelGGL((k_obstacle_gram<GTO_NB, GTO_OBS_DEEP_PD, false, true>), grid, dim3(256), lds, st, jobs_par, njobs_par, items_par, nitems_par, geo.nG, geo.m_nG, n_regular, B, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_chunks, h->d_scenes,
                       bpl, sp, t_begin, nT, fixed_mode, geo, 0);
  else if (hot && h->np == GTO_NB)
    hipLaunchKernelGGL((k_obstacle_gram<GTO_NB, GTO_OBS_MAIN_PD, false, true>), grid, dim3(256), lds, st, jobs_par, njobs_par, items_par, nitems_par, geo.nG, geo.m_nG, n_regular, B, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_chunks, h->d_scenes, bpl, sp,
                       t_begin, nT, fixed_mode, geo, 0);
  else if (hot)
    hipLaunchKernelGGL((k_obstacle_gram<16, GTO_OBS_MAIN_PD, false, true>), grid, dim3(256), lds, st, jobs_par, njobs_par, items_par, nitems_par, geo.nG, geo.m_nG, n_regular, B, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_chunks, h->d_scenes, bpl, sp,
                       t_begin, nT, fixed_mode, geo, 0);
  else if (h->np == GTO_NB && deep)  // few instances in flight: the variant that keeps a wave's record gathers in flight together
    hipLaunchKernelGGL((k_obstacle_gram<GTO_NB, GTO_OBS_DEEP_PD>), grid, dim3(256), lds, st, jobs_par, njobs_par, items_par, nitems_par, geo.nG, geo.m_nG, n_regular, B, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_chunks, h->d_scenes,
                       bpl, sp, t_begin, nT, fixed_mode, geo, 0);
  else if (h->np == GTO_NB)
    hipLaunchKernelGGL(k_obstacle_gram<GTO_NB>, grid, dim3(256), lds, st, jobs_par, njobs_par, items_par, nitems_par, geo.nG, geo.m_nG, n_regular, B, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_chunks, h->d_scenes, bpl, sp,
                       t_begin, nT, fixed_mode, geo, 0);
  else
    hipLaunchKernelGGL(k_obstacle_gram<16>, grid, dim3(256), lds, st, jobs_par, njobs_par, items_par, nitems_par, geo.nG, geo.m_nG, n_regular, B, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_chunks, h->d_scenes, bpl, sp,
                       t_begin, nT, fixed_mode, geo, 0);
  if (sweep) {  // the crew: items n_regular, n_regular + 1, ... of the list, if there are any
    hipLaunchKernelGGL((k_obstacle_gram<GTO_NB, GTO_OBS_MAIN_PD, true>), dim3(GTO_SWEEP_WGS), dim3(256), lds, st, jobs_par, njobs_par, items_par, nitems_par, geo.nG, geo.m_nG, GTO_SWEEP_WGS, B, h->d_rb, h->d_px, h->d_py, h->d_pz,
                       h->d_chunks, h->d_scenes, bpl, sp, t_begin, nT, fixed_mode, geo, n_regular);
  }
  if (timed) return prof_end(h, st);
  return GTO_OK;
}

static int check_scene_ids_host(gto_handle* h, const int32_t* ids, int B) {
  for (int b = 0; b < B; ++b)
    if (ids[b] < 0 || (size_t)ids[b] >= h->scenes.size() || !h->scenes[ids[b]].valid)
      return fail(h, GTO_ERR_NO_SCENE, "scene_id refers to a scene that was never set");
  for (int b = 0; b < B; ++b)
    if (!h->scenes[ids[b]].r_all)
      return fail(h, GTO_ERR_NO_SCENE, "scene_id refers to a values-only scene (gto_set_scene_values): it serves gto_plan_cost and gto_eval_points only");
  return GTO_OK;
}

}  // extern "C"

extern "C" {

int gto_solve_batch_device(gto_handle* h, int32_t B, int32_t n_max, const int32_t* scene_id, const double* qc,
                           const double* goals, const int32_t* n_goals, const double* standoff, const double* base_pos,
                           const double* Q0, double* Q_out, double* dQ_out, double* cost_out, int32_t* iters_out,
                           int32_t* status_out, void* stream) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (B < 0 || n_max < 1) return fail(h, GTO_ERR_INVALID_ARG, "B must be >= 0 and n_max >= 1");
  if (B == 0) return GTO_OK;
  if (B > GTO_JOB_MASK) return fail(h, GTO_ERR_UNSUPPORTED, "more than 16.7 million instances in one call");
  if (!scene_id || !qc || !goals || !n_goals || !base_pos || !Q0) return fail(h, GTO_ERR_INVALID_ARG, "null input array");
  if (h->scenes.empty()) return fail(h, GTO_ERR_NO_SCENE, "no scene has been set");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  int rc = ensure_workspace(h, B);
  if (rc) return rc;
  SolveParams sp = make_params(h, n_max, standoff != nullptr);
  BatchPtrs bp = make_ptrs(h, scene_id, qc, goals, n_goals, standoff, base_pos, Q0);
  const int T = sp.T;
  h->last_launches = 0;
  h->last_ms = 0.0;
  if (h->dbg) HIPCHK(h, hipMemsetAsync(h->dbg + 40, 0, 216 * sizeof(long long), st));

  // At most W instances per lane are in flight; the step kernel of an instance that finishes puts the next one of its lane
  // that has not started into the next round's live list, so every round works on a full house until the lane runs out,
  // instead of dragging the tail of its slowest instances through ever emptier rounds.
  //
  // LANES.  A call's instances are dealt to up to h->lanes_max lanes (contiguous ranges of at least h->lane_min), each
  // with a stream and lists of its own over the ONE workspace of the call (everything about an instance is indexed by its
  // id in the batch): one lane's obstacle launch overlaps another's step launch while the GPU is full.  One host thread
  // feeds all of them round-robin.  Towards the end of the call every lane is down to a handful of stragglers whose
  // iterations are one dependent round each; four such chains sharing the command processor advance at 35-55 us a round
  // where one alone takes 21-27, so a lane whose remaining instances are all in flight and at most h->adopt_below hands
  // them to lane 0 (k_adopt), which then runs ONE chain for everybody.
  const int L = std::max(1, std::min(std::min(h->lanes_max, GTO_MAX_LANES), B / std::max(1, h->lane_min)));
  const int kcap = sp.kcap, nF = h->rb.n_frames;
  struct LaneCtx {
    hipStream_t st = nullptr;
    BatchPtrs bp;
    SolveParams sp;
    int lo = 0, n = 0, W = 0, cap = 0;  // first instance, instances, in flight at the start, list capacity
    int room = 0;                       // positions its lists can have in use: W + what it adopted
    int span_prev = 0;                  // static positions: positions the current lists span (the last one-candidate launch's grid), 0: compact lists
    int n_resp = 0;                     // instances whose end this lane's finished-counter counts (own + adopted)
    int k = 0, known_done = 0, seen_round = -1, k_prev = 1;
    bool items_ready = false, pb_off = false, handed = false;
    double ratio_max = 0.0;  // items per job, the largest the lane's rounds published
    long end_us = 0, few_us = 0;  // (GTO_LANE_DEBUG) when the lane's thread returned / enqueued its first few-instance round, from the start of the threads
    unsigned long long* h_prog = nullptr;
  };
  std::vector<LaneCtx> lanes(L);
  const int adopt_room = L > 1 ? (L - 1) * std::max(0, h->adopt_below) : 0;
  size_t off_live[GTO_MAX_LANES + 1] = {0}, off_qfs[GTO_MAX_LANES + 1] = {0}, off_items[GTO_MAX_LANES + 1] = {0}, lane_zws[GTO_MAX_LANES + 1] = {0};
  for (int l = 0; l < L; ++l) {
    LaneCtx& ln = lanes[l];
    ln.lo = (int)((long long)B * l / L);
    ln.n = (int)((long long)B * (l + 1) / L) - ln.lo;
    ln.W = std::min(ln.n, h->slots);
    ln.cap = ln.W + (l == 0 ? adopt_room : 0);
    ln.n_resp = ln.n;
    ln.room = ln.W;
    off_live[l + 1] = off_live[l] + 2 * (size_t)(1 + kcap) * ln.cap + 32;
    off_qfs[l + 1] = off_qfs[l] + 2 * (size_t)ln.cap * kcap * T * nF;
    off_items[l + 1] = off_items[l] + 2 * ((size_t)ln.cap * kcap * (T - 2) + GTO_ITEM_SLACK);
    lane_zws[l + 1] = lane_zws[l] + (size_t)ln.cap;
  }
  if ((rc = ensure(h, h->livebuf, off_live[L] * sizeof(int32_t)))) return rc;
  if ((rc = ensure(h, h->qfs, off_qfs[L] * sizeof(double)))) return rc;
  if ((rc = ensure(h, h->itembuf, off_items[L] * sizeof(int2)))) return rc;
  if (h->np != GTO_NB && (rc = ensure(h, h->zws, lane_zws[L] * (size_t)(T - 2) * h->np * h->np * sizeof(double)))) return rc;
  for (int l = 0; l < L; ++l) {
    LaneCtx& ln = lanes[l];
    ln.sp = sp;
    ln.bp = bp;
    ln.bp.live = (int32_t*)h->livebuf.p + off_live[l];
    ln.bp.jobs = ln.bp.live + 2 * ln.cap;
    ln.bp.nlive = ln.bp.jobs + 2 * ln.cap * kcap;
    ln.bp.next = ln.bp.nlive + 4;
    ln.bp.qfs = (double*)h->qfs.p + off_qfs[l];
    ln.bp.items = h->np == GTO_NB ? (int2*)h->itembuf.p + off_items[l] : nullptr;
    ln.bp.n_done = (int32_t*)h->ndone.p + 16 * l;
    ln.bp.cap = ln.cap;
    ln.bp.b0 = ln.lo;
    ln.bp.w0 = ln.W;
    ln.bp.n_total = ln.lo + ln.n;
    ln.h_prog = h->h_progress + 8 * l;
    ln.bp.progress = h->d_progress + 8 * l;
    // One lane: the caller's stream.  Several: streams of the handle's own, created one after the other -- the runtime deals
    // streams to its few hardware queues (GPU_MAX_HW_QUEUES, default 4) in the order of their creation, and two lanes on
    // one queue run one after the other (the caller's stream next to three new ones: 126 k trajectories/s instead of 225 k);
    // the caller's stream carries the start and the end of the call and waits in between.
    if (L == 1) ln.st = st;
    else if (l < h->n_user_lane_streams) ln.st = h->user_lane_stream[l];  // gto_set_lane_streams
    else {
      if (!h->lane_stream[l]) {
        // Streams of the greatest priority by default: the runtime gives them hardware queues of their own, one per stream
        // in the order of creation, so four lanes sit behind four dispatcher pipes.  Streams of the default priority share
        // the process's four queues with every other stream it has created, and two lanes whose queues sit behind one pipe
        // split its workgroup dispatch rate (the evaluation launch, five thousand mostly empty workgroups, 75 us instead
        // of 48; rocprofv3 queue ids 1 and 5 in gpurun_out traces of round 5): 118 k instead of 196 k trajectories/s for
        // one call of 1280 instances.  GTO_LANE_PRIO=0 default priority, 2 least.
        static const int prio_mode = getenv("GTO_LANE_PRIO") ? atoi(getenv("GTO_LANE_PRIO")) : 1;
        int lo_p = 0, hi_p = 0;
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));  // (least, greatest): greatest is the smaller number
        if (prio_mode == 1) HIPCHK(h, hipStreamCreateWithPriority(&h->lane_stream[l], hipStreamNonBlocking, hi_p));
        else if (prio_mode == 2) HIPCHK(h, hipStreamCreateWithPriority(&h->lane_stream[l], hipStreamNonBlocking, lo_p));
        else HIPCHK(h, hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking));
      }
      ln.st = h->lane_stream[l];
    }
    if (!h->lane_event[l]) HIPCHK(h, hipEventCreateWithFlags(&h->lane_event[l], hipEventDisableTiming));
  }
  HIPCHK(h, hipMemsetAsync(h->ndone.p, 0, 16 * sizeof(int32_t) * L, st));
  // the finished-counter and the round counter reach the host through words in pinned memory that the step kernel
  // writes; the tag tells this call's values from what the last launches of the previous call may still be writing
  h->progress_tag = h->progress_tag + 1 ? h->progress_tag + 1 : 1;
  for (int l = 0; l < L; ++l) lanes[l].bp.progress_tag = (unsigned long long)h->progress_tag << 32;
  if (h->profiling) {
    if ((rc = ensure(h, h->counters, GTO_PROF_VARIANTS * 64 * sizeof(unsigned long long)))) return rc;
    HIPCHK(h, hipMemsetAsync(h->counters.p, 0, GTO_PROF_VARIANTS * 64 * sizeof(unsigned long long), st));
    bp.work = (unsigned long long*)h->counters.p;  // 64 cells per kernel variant (launch_obstacle picks the variant's)
    for (int l = 0; l < L; ++l) lanes[l].bp.work = bp.work;
  }
  // The broad phase ahead of the obstacle launch, in the rounds that fill the GPU: the step kernel settles the waypoint
  // groups none of whose bounding spheres can reach a non-zero voxel record and lists the others (prebroad_tail); the
  // launch is laid out over that list.  Needs: the groups of the launch it feeds (consecutive waypoints), room for one
  // pass in the step kernel's dead LDS, at most GTO_PB_PARK parked frames in its serial walk over the kinematic tree.
  const int pb_tg = std::max(1, std::min(h->obs_tg, T - 2)), pb_ng = (T - 2 + pb_tg - 1) / pb_tg;
  const PbLayout pbl(T, h->rb.n_frames, h->pb_C, h->rb.pb_npar);
  const bool pb_able = h->prebroad && h->np == GTO_NB && pb_ng <= 64 && h->obs_interleave != 1 && h->rb.n_xst <= GTO_PB_PARK && pbl.pw >= 1 && h->pb_C >= 1;
  for (int l = 0; l < L; ++l) {
    SolveParams& lsp = lanes[l].sp;
    lsp.pb_tg = pb_tg, lsp.pb_ng = pb_ng, lsp.pb_pw = std::max(1, pbl.pw), lsp.pb_tab0 = pbl.tab0;
    lsp.pb_verify = h->dbg_cut == 10;
  }
  sp.pb_tg = pb_tg, sp.pb_ng = pb_ng, sp.pb_pw = std::max(1, pbl.pw), sp.pb_tab0 = pbl.tab0;
  for (int l = 0; l < L; ++l) {  // seeds, goal terms of the seeds, the lane's lists of round 0
    if (h->np == GTO_NB) hipLaunchKernelGGL(k_lm_init<GTO_NB>, dim3(lanes[l].n), dim3(256), 0, st, h->d_rb, lanes[l].bp, lanes[l].sp, B, 0);
    else hipLaunchKernelGGL(k_lm_init<16>, dim3(lanes[l].n), dim3(256), 0, st, h->d_rb, lanes[l].bp, lanes[l].sp, B, 0);
  }
  {
    BatchPtrs bpi = bp;  // the init pass indexes the batch directly
    if ((rc = launch_obstacle(h, st, bpi, sp, B, 0, 4, 1, false))) return rc;
  }
  if (L > 1) {
    HIPCHK(h, hipEventRecord(h->lane_event[0], st));
    for (int l = 0; l < L; ++l) HIPCHK(h, hipStreamWaitEvent(lanes[l].st, h->lane_event[0], 0));
  }
  // one round = evaluate the candidate trial trajectories of the instances in flight (obstacle kernel) +
  // accept/solve/new candidates (step kernel).  An instance may start late: enough rounds for every position to serve its
  // share one after the other.
  auto read_progress = [&](LaneCtx& ln) {
    const unsigned long long p0 = __atomic_load_n(ln.h_prog, __ATOMIC_RELAXED), p1 = __atomic_load_n(ln.h_prog + 1, __ATOMIC_RELAXED);
    if ((unsigned)(p0 >> 32) == h->progress_tag) ln.known_done = std::max(ln.known_done, (int)(p0 & 0xffffffffull));
    if ((unsigned)(p1 >> 32) == h->progress_tag) ln.seen_round = std::max(ln.seen_round, (int)(p1 & 0xffffffffull));
  };
  // naps of the throttle below: tens of microseconds, which the default timer slack of a thread (50 us) would double
  const int old_slack = prctl(PR_GET_TIMERSLACK);
  if (old_slack > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000UL);
  int rc_loop = GTO_OK;
  const auto tp_start = std::chrono::steady_clock::now();
  // one round of lane ln: returns GTO_OK or an error
  auto enqueue_round = [&](LaneCtx& ln, int lane_index) -> int {
    const int k = ln.k;
    const int in_flight = std::min(ln.room, ln.n_resp - ln.known_done);
    const bool few = in_flight <= h->few_instances;
    // positions this round's lists span: compact lists hold the instances in flight; lists of a launch with static positions
    // keep that launch's span (instances that finished without a successor left void positions behind)
    const int span = ln.span_prev ? ln.span_prev : in_flight;
    ln.span_prev = 0;
    if (few && !ln.few_us) ln.few_us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tp_start).count();
    const bool large_call = ln.n_resp > h->spec_deep;  // (its tail: the other lanes' launches are on the GPU, too)
    const int tg = few ? (large_call ? h->obs_tg_few_tail : h->obs_tg_few) : h->obs_tg;
    SolveParams& lsp = ln.sp;
    lsp.interleave = h->obs_interleave == 1 || (h->obs_interleave == 2 && few);
    const bool itemized = ln.items_ready && !few;
    ln.items_ready = false;
    lsp.round = k;
    lsp.parity = k & 1;
    // the goal workgroups skip fresh instances themselves: k_lm_init already produced the seed's goal terms
    lsp.k_eval = ln.k_prev;
    int rc_;
    // length of this round's item list, estimated: what the lane's step launches published last (a few rounds old; the
    // list changes by a few per cent a round), or, until then, the previous call's ratio of items to jobs
    int items_hint = 0;
    if (itemized && h->item_grid) {
      const unsigned long long p2 = __atomic_load_n(ln.h_prog + 2, __ATOMIC_RELAXED);
      const int jobs_bound = in_flight * ln.k_prev;
      if ((unsigned)(p2 >> 32) == h->progress_tag && (p2 & 0xfffffull) > 0 && (p2 & 0xfffffull) < 0xfffffull) {
        const double items_seen = (double)(p2 & 0xfffffull), jobs_seen = (double)((p2 >> 20) & 0xfffull);
        items_hint = (int)(1.5 * items_seen) + 256;
        if (jobs_seen > 0 && jobs_seen < 4095) ln.ratio_max = std::max(ln.ratio_max, items_seen / jobs_seen);
      } else if (h->items_per_job_prior > 0.0) {
        items_hint = (int)(1.25 * h->items_per_job_prior * jobs_bound) + 256;
      }
      if (h->item_hint_forced > 0) items_hint = h->item_hint_forced;  // (tests: a launch of a few workgroups, the crew does the rest)
    }
    if ((rc_ = launch_obstacle(h, ln.st, ln.bp, lsp, B, 2, T - 2, 0, h->profiling, true, span * ln.k_prev, tg, few && h->obs_deep && in_flight <= h->obs_deep_max, itemized, items_hint))) return rc_;
    if (h->np == GTO_NB) {
      if (few && h->step_nw_few == 8) {
        // few instances in flight: eight waves per instance and candidate trial points ahead of their evaluation
        const int k_budget = std::max(1, h->spec_jobs / std::max(1, in_flight));
        lsp.k_acc = in_flight <= std::min(h->spec_deep, h->spec_few) ? std::min(std::min(large_call ? h->spec_acc_tail : h->spec_acc, k_budget), h->spec_kmax) : 1;
        lsp.spec_streak = large_call ? h->spec_streak_tail : h->spec_streak;
        lsp.k_rej = in_flight <= h->spec_few ? std::min(std::max(std::min(h->spec_rej, k_budget), h->spec_rej_few), h->spec_kmax) : 1;
        const int kl = std::max(lsp.k_acc, lsp.k_rej);
        lsp.pb_next = 0;
        if (h->profiling && (rc_ = prof_begin(h, ln.st, GTO_PROF_STEP_FEW, in_flight))) return rc_;
        lsp.static_pos = 0;
        hipLaunchKernelGGL((k_lm_step<8, GTO_KSPEC>), dim3(span), dim3(512), lm_lds_bytes(T, kl), ln.st, h->d_rb, h->d_pbchunks, ln.bp, lsp, B);
        if (h->profiling && (rc_ = prof_end(h, ln.st))) return rc_;
        ln.k_prev = kl;
      } else {
        lsp.k_acc = lsp.k_rej = 1;
        const bool pb_ok = pb_able && std::min(ln.W, ln.n) > h->few_instances;
        // a call in which the broad phase settles next to nothing (a robot inside a shelf) stops running it: the last
        // itemized round the host has seen listed more than 1 - GTO_PB_MIN_GAIN of its (job, group) pairs
        if (pb_ok && !ln.pb_off && !lsp.pb_verify) {
          const unsigned long long p2 = __atomic_load_n(ln.h_prog + 2, __ATOMIC_RELAXED);
          if ((unsigned)(p2 >> 32) == h->progress_tag) {
            const double jobs_seen = (double)((p2 >> 20) & 0xfffull), items_seen = (double)(p2 & 0xfffffull);
            if (jobs_seen > 0 && items_seen > 0 && jobs_seen < 4095 && k >= 12 && items_seen > (1.0 - h->pb_min_gain) * jobs_seen * pb_ng) ln.pb_off = true;
          }
        }
        lsp.pb_next = pb_ok && !ln.pb_off && !few;
        if (h->profiling && (rc_ = prof_begin(h, ln.st, GTO_PROF_STEP, in_flight))) return rc_;
        lsp.static_pos = h->static_pos && L == 1 && !few;
        hipLaunchKernelGGL((k_lm_step<4, 1>), dim3(span), dim3(256), h->lm_lds, ln.st, h->d_rb, h->d_pbchunks, ln.bp, lsp, B);
        if (lsp.static_pos) ln.span_prev = span;
        if (h->profiling && (rc_ = prof_end(h, ln.st))) return rc_;
        ln.k_prev = 1;
        ln.items_ready = lsp.pb_next != 0;
      }
    } else {
      if (h->profiling && (rc_ = prof_begin(h, ln.st, GTO_PROF_STEP, in_flight))) return rc_;
      hipLaunchKernelGGL(k_lm_step_wide<16>, dim3(in_flight), dim3(GTO_WIDE_NT), h->lm_lds, ln.st, h->d_rb, ln.bp, lsp, B,
                         (double*)h->zws.p + (size_t)lane_zws[lane_index] * (T - 2) * h->np * h->np);
      if (h->profiling && (rc_ = prof_end(h, ln.st))) return rc_;
    }
    ln.k++;
    return GTO_OK;
  };
  // One host thread per lane (lane 0: the caller's), as many as the call has lanes: a round of a lane with few instances
  // in flight lasts 25-50 us and takes two launches to enqueue, which one thread cannot do for four lanes.
  struct Handover { int lane, parity, count, k_prev; };
  std::mutex mu;                     // hand-over requests, the error string, the profiling records
  std::vector<Handover> requests;
  int reserved = 0;                  // instances granted to lane 0 whose k_adopt it has not enqueued yet
  std::atomic<int> rem0_pub{lanes[0].n}, others_open{L - 1}, failed{0};
  const int adopt_limit = std::min(h->few_instances, lanes[0].cap);
  h->prof_mu = L > 1 ? &mu : nullptr;
  // waits until lane ln may enqueue its next round (throttle: never more than `ahead` rounds in front of the last step
  // launch seen running, so that the launches stay sized to what is left and the empty rounds after the last instance
  // finishes stay few; sleep-poll, not a blocking wait: the runtime spins in those, one host core per lane)
  auto throttle = [&](LaneCtx& ln) -> int {
    for (long naps = 0;; ++naps) {
      read_progress(ln);
      const bool few = std::min(ln.room, ln.n_resp - ln.known_done) <= h->few_instances;
      if (ln.k - ln.seen_round <= (few ? h->ahead_few : h->ahead) + 1 || failed.load(std::memory_order_relaxed)) return GTO_OK;
      std::this_thread::sleep_for(std::chrono::microseconds(few ? h->nap_few_us : h->nap_us));
      if ((naps & 1023) == 1023) {  // a stream that went idle or failed without reaching the round: do not wait for ever
        const hipError_t qe = hipStreamQuery(ln.st);
        if (qe != hipErrorNotReady) {
          read_progress(ln);
          if (ln.k - ln.seen_round > (few ? h->ahead_few : h->ahead) + 1) {
            std::lock_guard<std::mutex> g(mu);
            h->err = qe == hipSuccess ? "solve loop: the stream went idle before the rounds it was given ran" : std::string("solve loop: ") + hipGetErrorString(qe);
            return GTO_ERR_HIP;
          }
        }
      }
    }
  };
  auto lane_main = [&](int l) {
    LaneCtx& ln = lanes[l];
    (void)hipSetDevice(h->device);
    const int old_slack_ = prctl(PR_GET_TIMERSLACK);
    if (l > 0 && old_slack_ > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000UL);
    int rc_ = GTO_OK;
    for (;;) {
      if (failed.load(std::memory_order_relaxed)) break;
      if ((rc_ = throttle(ln))) break;
      int remaining = ln.n_resp - ln.known_done;
      if (l == 0) {
        // hand-overs granted since the last round: behind the lane's last launch (its event) and behind this lane's
        {
          std::lock_guard<std::mutex> g(mu);
          for (const Handover& r : requests) {
            LaneCtx& src = lanes[r.lane];
            if (hipStreamWaitEvent(ln.st, h->lane_event[r.lane], 0) != hipSuccess) { h->err = "solve loop: hand-over between lanes failed"; rc_ = GTO_ERR_HIP; break; }
            hipLaunchKernelGGL(k_adopt, dim3(1), dim3(256), 0, ln.st, src.bp, r.parity, ln.bp, ln.k & 1, kcap, T * nF, src.n_resp, r.count);
            ln.n_resp += r.count;
            ln.room = std::min(ln.cap, ln.room + r.count);
            ln.k_prev = std::max(ln.k_prev, r.k_prev);
            ln.items_ready = false;
            reserved -= r.count;
          }
          requests.clear();
          remaining = ln.n_resp - ln.known_done;
          rem0_pub.store(std::max(0, remaining), std::memory_order_relaxed);
        }
        if (rc_) break;
        if (remaining <= 0) {
          if (others_open.load(std::memory_order_acquire) == 0) {
            std::lock_guard<std::mutex> g(mu);
            if (requests.empty()) break;
            continue;
          }
          std::this_thread::sleep_for(std::chrono::microseconds(10));
          continue;
        }
      } else {
        if (remaining <= 0) break;
        // hand-over: everything this lane has left is in flight (remaining <= W) and few, and lane 0 has room
        if (h->adopt_below > 0 && remaining <= h->adopt_below && remaining <= ln.W && ln.k > 0) {
          std::lock_guard<std::mutex> g(mu);
          const int rem0 = rem0_pub.load(std::memory_order_relaxed);
          if (rem0 <= lanes[0].W && rem0 + reserved + remaining <= adopt_limit) {
            if (hipEventRecord(h->lane_event[l], ln.st) != hipSuccess) { h->err = "solve loop: hand-over between lanes failed"; rc_ = GTO_ERR_HIP; break; }
            reserved += remaining;
            requests.push_back({l, ln.k & 1, remaining, ln.k_prev});
            ln.handed = true;
            break;
          }
        }
      }
      const int max_rounds = ((ln.n_resp + ln.W - 1) / std::max(1, ln.W) + 1) * (sp.max_iter + 2) + 8;
      if (ln.k > max_rounds) {
        std::lock_guard<std::mutex> g(mu);
        h->err = "solve loop: a lane ran out of rounds";
        rc_ = GTO_ERR_HIP;
        break;
      }
      if ((rc_ = enqueue_round(ln, l))) break;
    }
    if (rc_) failed.store(rc_, std::memory_order_relaxed);
    ln.end_us = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tp_start).count();
    if (l > 0) {
      others_open.fetch_sub(1, std::memory_order_release);
      if (old_slack_ > 1000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old_slack_);
    }
  };
  {
    static const bool lane_dbg = getenv("GTO_LANE_DEBUG") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    std::vector<std::thread> workers;
    for (int l = 1; l < L; ++l) workers.emplace_back(lane_main, l);
    const auto tp1 = std::chrono::steady_clock::now();
    lane_main(0);
    const auto tp2 = std::chrono::steady_clock::now();
    for (auto& w : workers) w.join();
    const auto tp3 = std::chrono::steady_clock::now();
    if (lane_dbg) {
      auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
      fprintf(stderr, "[gto lanes] B %d L %d: threads started %ld us | lane 0 returned %ld us | joined %ld us | rounds", B, L, us(tp0, tp1), us(tp0, tp2), us(tp0, tp3));
      for (int l = 0; l < L; ++l) fprintf(stderr, " %d%s@%ld(few@%ld)", lanes[l].k, lanes[l].handed ? "h" : "", lanes[l].end_us, lanes[l].few_us);
      fprintf(stderr, "\n");
    }
  }
  h->prof_mu = nullptr;
  rc_loop = failed.load();
  {
    double r_ = 0.0;
    for (int l = 0; l < L; ++l) r_ = std::max(r_, lanes[l].ratio_max);
    if (r_ > 0.0) h->items_per_job_prior = r_;
  }
  // the other lanes' work is behind the finalisation on the caller's stream
  for (int l = 0; l < L && L > 1 && !rc_loop; ++l)
    if (!lanes[l].handed) {
      if (hipEventRecord(h->lane_event[l], lanes[l].st) != hipSuccess || hipStreamWaitEvent(st, h->lane_event[l], 0) != hipSuccess) {
        h->err = "solve loop: joining the lanes failed";
        rc_loop = GTO_ERR_HIP;
      }
    }
  if (rc_loop)  // leave nothing of this call running on streams the caller does not know about
    for (int l = 0; l < L && L > 1; ++l) (void)hipStreamSynchronize(lanes[l].st);
  bp.live = lanes[0].bp.live, bp.cap = lanes[0].bp.cap;  // (what the debug print below and k_lm_finalize see: lists are not read there)
  if (old_slack > 1000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)old_slack);
  if (rc_loop) return rc_loop;
  hipLaunchKernelGGL(k_lm_finalize, dim3(B), dim3(64), 0, st, h->d_rb, bp, sp, B, Q_out, dQ_out, cost_out, iters_out, status_out);
  HIPCHK(h, hipGetLastError());
  if (h->dbg) {
    HIPCHK(h, hipStreamSynchronize(st));
    long long t[256];
    HIPCHK(h, hipMemcpy(t, h->dbg, sizeof t, hipMemcpyDeviceToHost));
    fprintf(stderr, "[gto dbg] step-kernel phases (cycles) P0+P1 %lld | P2 %lld | diag %lld | dense %lld | back %lld | P4 %lld | P5 %lld | s_dense %lld\n",
            t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[9]);
    if (h->np != GTO_NB)
      fprintf(stderr, "[gto dbg] wide step kernel, sweeps (cycles): downward: %lld in %lld dense inversions, sweep %lld | upward: %lld in %lld dense inversions, sweep %lld\n",
              t[56], t[57], t[58], t[59], t[60], t[61]);
    fprintf(stderr, "[gto dbg] solve by wave (cycles from the start of P3): downward sweep: diagonal stretch %lld, its dense blocks %lld | upward sweep %lld | meeting block %lld | back substitution from its start: downward wave dense %lld, diagonal stretch %lld | upward wave %lld\n",
            t[3] - t[2], t[41] - t[2], t[17] - t[2], t[42] - t[4], t[18] - t[42], t[19] - t[42], t[26] - t[42]);
    fprintf(stderr, "[gto dbg] sweeps of the other candidates' waves (2..6) done at (cycles from the start of P3, 0 = no candidate): %lld %lld %lld %lld %lld | barrier passed at %lld\n",
            t[43] ? t[43] - t[2] : 0, t[44] ? t[44] - t[2] : 0, t[45] ? t[45] - t[2] : 0, t[46] ? t[46] - t[2] : 0, t[47] ? t[47] - t[2] : 0, t[4] - t[2]);
    fprintf(stderr, "[gto dbg] broad phase in the step kernel's tail (cycles): entry+tables %lld | barrier %lld | A (local transforms, pass 0) %lld | B (chain) %lld | C (sphere tests) %lld | other passes %lld | outputs %lld\n",
            t[33] - t[6], t[34] - t[33], t[35] - t[34], t[36] - t[35], t[37] - t[36], t[38] - t[37], t[39] - t[38]);
    fprintf(stderr, "[gto dbg] P2 split (cycles): loads+barrier %lld | b-vector+masks %lld | blocks %lld | e,y+barrier %lld\n", t[28] - t[1], t[29] - t[28], t[30] - t[29], t[2] - t[30]);
    fprintf(stderr, "[gto dbg] fk_mfma_tree (cycles): local %lld | rounds %lld %lld %lld %lld | outputs %lld\n", t[21] - t[20], t[22] - t[21], t[23] - t[22], t[24] - t[23], t[25] - t[24], t[27] - t[25]);
    // (only with -DGTO_DEBUG_LONGEST_WG: the extra clocks cost the tuned obstacle kernel registers)
    fprintf(stderr, "[gto dbg] longest regular obstacle workgroup of the call: %lld cycles with %lld surviving chunks | goal-term wavefront of instance 0: %lld cycles\n",
            t[40] >> 16, t[40] & 0xffff, t[32] - t[31]);
    if (t[64] || t[65]) {  // -DGTO_DEBUG_LONGEST_WG
      long long tot_ = 0;
      for (int i = 64; i < 128; ++i) tot_ += t[i];
      fprintf(stderr, "[gto dbg] regular obstacle workgroups of the call: %lld, none of whose keys got a contribution: %lld | surviving chunks per workgroup (0,1,2,...,63+):", tot_, t[48]);
      for (int i = 64; i < 128; ++i) fprintf(stderr, " %lld", t[i]);
      fprintf(stderr, "\n");
      fprintf(stderr, "[gto dbg] ticks those workgroups ran, summed by surviving chunks (0,1,2,...,63+):");
      for (int i = 192; i < 256; ++i) fprintf(stderr, " %lld", t[i]);
      fprintf(stderr, "\n");
    }
    fprintf(stderr, "[gto dbg] broad phase of the step kernel (GTO_DEBUG_CUT=10, -DGTO_DEBUG_LONGEST_WG: settled groups are looked at anyway): %lld groups settled, %lld of them with a surviving chunk, %lld with a CONTRIBUTION (must be 0)\n", t[49], t[50], t[55]);
    {
      fprintf(stderr, "[gto dbg] workgroups without a surviving chunk by the index shift their closest chunk tolerates (0,1,2,...,63+):");
      for (int i = 128; i < 192; ++i) fprintf(stderr, " %lld", t[i]);
      fprintf(stderr, "\n");
    }
    fprintf(stderr, "[gto dbg] obstacle WG (b=0,t=T-1) cycles: prologue %lld | broad %lld | loop %lld | epilogue %lld | active chunks %lld | prologue up to the chain %lld\n",
            t[11] - t[10], t[12] - t[11], t[13] - t[12], t[14] - t[13], t[15], t[16] - t[10]);
  }
  if (h->profiling) {
    HIPCHK(h, hipStreamSynchronize(st));
    for (int v = 0; v < GTO_PROF_VARIANTS; ++v) h->prof_ms[v] = 0.0, h->prof_launches[v] = h->prof_wgs[v] = 0, h->prof_points[v] = 0;
    int n_obs = 0;
    for (int i = 0; i < h->last_launches; ++i) {
      float ms = 0.f;
      HIPCHK(h, hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
      const int v = h->ev_variant[i];
      h->prof_ms[v] += ms, h->prof_launches[v] += 1, h->prof_wgs[v] += h->ev_wgs[i];
      n_obs += v == GTO_PROF_OBSTACLE || v == GTO_PROF_OBSTACLE_FEW;
    }
    unsigned long long cells[GTO_PROF_VARIANTS * 64];
    HIPCHK(h, hipMemcpy(cells, bp.work, sizeof cells, hipMemcpyDeviceToHost));
    for (int v = 0; v < GTO_PROF_VARIANTS; ++v)
      for (int c = 0; c < 64; ++c) h->prof_points[v] += cells[64 * v + c];
    // gto_last_kernel_time / _work: the obstacle kernel, both variants together (what they always reported)
    h->last_ms = h->prof_ms[GTO_PROF_OBSTACLE] + h->prof_ms[GTO_PROF_OBSTACLE_FEW];
    h->last_launches = n_obs;
    h->last_counters[0] = h->prof_points[GTO_PROF_OBSTACLE] + h->prof_points[GTO_PROF_OBSTACLE_FEW];
    h->last_counters[1] = 0;
  }
  return GTO_OK;
}

// host-pointer staging helpers
static int ensure_pinned(gto_handle* h, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return GTO_OK;
  if (b.p) HIPCHK(h, hipHostFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  const size_t want = bytes + bytes / 4 + 256;
  HIPCHK(h, hipHostMalloc(&b.p, want));
  b.cap = want;
  return GTO_OK;
}
static int stage_in(gto_handle* h, int slot, const void* src, size_t bytes, const void** dptr) {
  *dptr = nullptr;
  if (!src) return GTO_OK;
  int rc = ensure(h, h->in[slot], bytes);
  if (rc) return rc;
  if ((rc = ensure_pinned(h, h->pin_in[slot], bytes))) return rc;  // free again: every entry point ends with a stream sync
  memcpy(h->pin_in[slot].p, src, bytes);
  HIPCHK(h, hipMemcpyAsync(h->in[slot].p, h->pin_in[slot].p, bytes, hipMemcpyHostToDevice, h->stream));
  *dptr = h->in[slot].p;
  return GTO_OK;
}
static int stage_out(gto_handle* h, int slot, const void* host, size_t bytes, void** dptr) {
  *dptr = nullptr;
  if (!host) return GTO_OK;
  int rc = ensure(h, h->out[slot], bytes);
  if (rc) return rc;
  *dptr = h->out[slot].p;
  return GTO_OK;
}
static int fetch_out(gto_handle* h, int slot, void* host, size_t bytes) {
  if (!host) return GTO_OK;
  int rc = ensure_pinned(h, h->pin_out[slot], bytes);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->pin_out[slot].p, h->out[slot].p, bytes, hipMemcpyDeviceToHost, h->stream));
  h->pending_out.push_back({host, h->pin_out[slot].p, bytes});
  return GTO_OK;
}
// after the stream sync that follows the fetch_out calls of an entry point: pinned -> the caller's arrays
static int sync_and_finish_out(gto_handle* h) {
  const hipError_t e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) {
    h->pending_out.clear();
    HIPCHK(h, e);
  }
  for (const auto& po : h->pending_out) memcpy(po.host, po.pin, po.bytes);
  h->pending_out.clear();
  return GTO_OK;
}

int gto_solve_ik_batch(gto_handle* h, int32_t B, const int32_t* scene_id, const double* q0, const double* goals,
                       const double* base_pos, int32_t max_iter, double* q_out, double* cost_out, int32_t* iters_out,
                       int32_t* status_out) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (B < 0 || max_iter < 0) return fail(h, GTO_ERR_INVALID_ARG, "B and max_iter must be >= 0");
  if (B == 0) return GTO_OK;
  if (!q0 || !goals || !q_out) return fail(h, GTO_ERR_INVALID_ARG, "null input array");
  if (h->np != GTO_NB) return fail(h, GTO_ERR_UNSUPPORTED, "gto_solve_ik_batch handles up to eight optimised joints");
  int rc;
  if (scene_id && (rc = check_scene_ids_host(h, scene_id, B))) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t ndof = h->rb.ndof;
  std::vector<double> zeros;
  if (scene_id && !base_pos) {
    zeros.assign((size_t)B * 3, 0.0);
    base_pos = zeros.data();
  }
  const void *d_sid = nullptr, *d_q0, *d_goals, *d_base = nullptr;
  void *d_q, *d_cost, *d_it, *d_stat;
  if (scene_id && (rc = stage_in(h, 0, scene_id, B * sizeof(int32_t), &d_sid))) return rc;
  if ((rc = stage_in(h, 1, q0, B * ndof * sizeof(double), &d_q0))) return rc;
  if ((rc = stage_in(h, 2, goals, (size_t)B * 16 * sizeof(double), &d_goals))) return rc;
  if (scene_id && (rc = stage_in(h, 5, base_pos, (size_t)B * 3 * sizeof(double), &d_base))) return rc;
  if ((rc = stage_out(h, 0, q_out, B * ndof * sizeof(double), &d_q))) return rc;
  if ((rc = stage_out(h, 2, cost_out, B * sizeof(double), &d_cost))) return rc;
  if ((rc = stage_out(h, 3, iters_out, B * sizeof(int32_t), &d_it))) return rc;
  if ((rc = stage_out(h, 4, status_out, B * sizeof(int32_t), &d_stat))) return rc;
  SolveParams sp = make_params(h, 1, false);
  sp.max_iter = max_iter;
  const size_t lds = (size_t)ik_lds_doubles(h->rb.n_frames, h->rb.n_links, h->rb.n_opt) * sizeof(double);
  if (lds > 150 * 1024) return fail(h, GTO_ERR_UNSUPPORTED, "robot too large for the IK kernel's LDS");
  HIPCHK(h, raise_dynamic_lds((const void*)k_ik_solve, lds));
  hipLaunchKernelGGL(k_ik_solve, dim3(B), dim3(256), lds, h->stream, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_chunks,
                     h->d_scenes, (const int32_t*)d_sid, (const double*)d_q0, (const double*)d_goals, (const double*)d_base, sp,
                     B, (double*)d_q, (double*)d_cost, (int32_t*)d_it, (int32_t*)d_stat);
  HIPCHK(h, hipGetLastError());
  if ((rc = fetch_out(h, 0, q_out, B * ndof * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 2, cost_out, B * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 3, iters_out, B * sizeof(int32_t)))) return rc;
  if ((rc = fetch_out(h, 4, status_out, B * sizeof(int32_t)))) return rc;
  if ((rc = sync_and_finish_out(h))) return rc;
  return GTO_OK;
}

int gto_solve_base_batch(gto_handle* h, int32_t B, int32_t n_max, const int32_t* n_goals, const double* qc,
                         const double* goals, double effort_weight, int32_t max_iter, double* y_out, double* q_out,
                         double* cost_out, int32_t* iters_out, int32_t* status_out) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (B < 0 || max_iter < 0) return fail(h, GTO_ERR_INVALID_ARG, "B and max_iter must be >= 0");
  if (n_max < 1 || n_max > GTO_MAX_BASE_GOALS) return fail(h, GTO_ERR_UNSUPPORTED, "n_max must be in [1, 32]");
  if (h->np != GTO_NB) return fail(h, GTO_ERR_UNSUPPORTED, "gto_solve_base_batch handles up to eight optimised joints");
  if (B == 0) return GTO_OK;
  if (!n_goals || !qc || !goals || !y_out || !q_out) return fail(h, GTO_ERR_INVALID_ARG, "null input array");
  for (int b = 0; b < B; ++b)
    if (n_goals[b] < 1 || n_goals[b] > n_max) return fail(h, GTO_ERR_INVALID_ARG, "n_goals[b] must be in [1, n_max]");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t ndof = h->rb.ndof;
  int rc;
  const void *d_qc, *d_goals, *d_ng;
  void *d_q, *d_y, *d_cost, *d_it, *d_stat;
  if ((rc = stage_in(h, 1, qc, B * ndof * sizeof(double), &d_qc))) return rc;
  if ((rc = stage_in(h, 2, goals, (size_t)B * n_max * 16 * sizeof(double), &d_goals))) return rc;
  if ((rc = stage_in(h, 3, n_goals, B * sizeof(int32_t), &d_ng))) return rc;
  if ((rc = stage_out(h, 0, q_out, (size_t)B * n_max * ndof * sizeof(double), &d_q))) return rc;
  if ((rc = stage_out(h, 1, y_out, (size_t)B * 3 * sizeof(double), &d_y))) return rc;
  if ((rc = stage_out(h, 2, cost_out, B * sizeof(double), &d_cost))) return rc;
  if ((rc = stage_out(h, 3, iters_out, B * sizeof(int32_t), &d_it))) return rc;
  if ((rc = stage_out(h, 4, status_out, B * sizeof(int32_t), &d_stat))) return rc;
  SolveParams sp = make_params(h, 1, false);
  sp.max_iter = max_iter;
  const size_t lds = (size_t)base_lds_doubles(n_max) * sizeof(double);
  if (lds > 160 * 1024) return fail(h, GTO_ERR_UNSUPPORTED, "goal set too large for the base kernel's LDS");
  HIPCHK(h, raise_dynamic_lds((const void*)k_base_solve, lds));
  hipLaunchKernelGGL(k_base_solve, dim3(B), dim3(256), lds, h->stream, h->d_rb, (const double*)d_qc, (const double*)d_goals,
                     (const int32_t*)d_ng, sp, effort_weight, n_max, (double*)d_y, (double*)d_q, (double*)d_cost,
                     (int32_t*)d_it, (int32_t*)d_stat, (const double*)nullptr, (const double*)nullptr);
  HIPCHK(h, hipGetLastError());
  if ((rc = fetch_out(h, 0, q_out, (size_t)B * n_max * ndof * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 1, y_out, (size_t)B * 3 * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 2, cost_out, B * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 3, iters_out, B * sizeof(int32_t)))) return rc;
  if ((rc = fetch_out(h, 4, status_out, B * sizeof(int32_t)))) return rc;
  if ((rc = sync_and_finish_out(h))) return rc;
  return GTO_OK;
}

#ifdef GTO_DEBUG_BASE_TIMING
// (debug builds only: tools/base_stamps.py) phase stamps of k_base_solve since the last call, then cleared
int gto_debug_base_stamps(long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_base_dbg), 16 * sizeof(long long)) != hipSuccess) return GTO_ERR_HIP;
  long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_base_dbg), z, sizeof z) == hipSuccess ? GTO_OK : GTO_ERR_HIP;
}
#endif

int gto_eval_base_objective(gto_handle* h, int32_t B, int32_t n_max, const int32_t* n_goals, const double* y,
                            const double* q, const double* goals, double effort_weight, double* cost_out) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (B < 0) return fail(h, GTO_ERR_INVALID_ARG, "B must be >= 0");
  if (n_max < 1 || n_max > GTO_MAX_BASE_GOALS) return fail(h, GTO_ERR_UNSUPPORTED, "n_max must be in [1, 32]");
  if (h->np != GTO_NB) return fail(h, GTO_ERR_UNSUPPORTED, "gto_eval_base_objective handles up to eight optimised joints");
  if (B == 0) return GTO_OK;
  if (!n_goals || !y || !q || !goals || !cost_out) return fail(h, GTO_ERR_INVALID_ARG, "null input array");
  for (int b = 0; b < B; ++b)
    if (n_goals[b] < 1 || n_goals[b] > n_max) return fail(h, GTO_ERR_INVALID_ARG, "n_goals[b] must be in [1, n_max]");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t ndof = h->rb.ndof;
  int rc;
  // the parameter joints come from the first goal's configuration of every set
  std::vector<double> qc((size_t)B * ndof);
  for (int b = 0; b < B; ++b) std::copy(q + (size_t)b * n_max * ndof, q + (size_t)b * n_max * ndof + ndof, qc.begin() + (size_t)b * ndof);
  const void *d_qc, *d_goals, *d_ng, *d_y0, *d_q0;
  void *d_q, *d_y, *d_cost;
  if ((rc = stage_in(h, 1, qc.data(), B * ndof * sizeof(double), &d_qc))) return rc;
  if ((rc = stage_in(h, 2, goals, (size_t)B * n_max * 16 * sizeof(double), &d_goals))) return rc;
  if ((rc = stage_in(h, 3, n_goals, B * sizeof(int32_t), &d_ng))) return rc;
  if ((rc = stage_in(h, 4, y, (size_t)B * 3 * sizeof(double), &d_y0))) return rc;
  if ((rc = stage_in(h, 6, q, (size_t)B * n_max * ndof * sizeof(double), &d_q0))) return rc;
  if ((rc = ensure(h, h->out[0], (size_t)B * n_max * ndof * sizeof(double)))) return rc;
  if ((rc = ensure(h, h->out[1], (size_t)B * 3 * sizeof(double)))) return rc;
  d_q = h->out[0].p;
  d_y = h->out[1].p;
  if ((rc = stage_out(h, 2, cost_out, B * sizeof(double), &d_cost))) return rc;
  SolveParams sp = make_params(h, 1, false);
  sp.max_iter = 0;
  const size_t lds = (size_t)base_lds_doubles(n_max) * sizeof(double);
  if (lds > 160 * 1024) return fail(h, GTO_ERR_UNSUPPORTED, "goal set too large for the base kernel's LDS");
  HIPCHK(h, raise_dynamic_lds((const void*)k_base_solve, lds));
  hipLaunchKernelGGL(k_base_solve, dim3(B), dim3(256), lds, h->stream, h->d_rb, (const double*)d_qc, (const double*)d_goals,
                     (const int32_t*)d_ng, sp, effort_weight, n_max, (double*)d_y, (double*)d_q, (double*)d_cost,
                     (int32_t*)nullptr, (int32_t*)nullptr, (const double*)d_y0, (const double*)d_q0);
  HIPCHK(h, hipGetLastError());
  if ((rc = fetch_out(h, 2, cost_out, B * sizeof(double)))) return rc;
  if ((rc = sync_and_finish_out(h))) return rc;
  return GTO_OK;
}

int gto_solve_batch(gto_handle* h, int32_t B, int32_t n_max, const int32_t* scene_id, const double* qc, const double* goals,
                    const int32_t* n_goals, const double* standoff, const double* base_pos, const double* Q0,
                    double* Q_out, double* dQ_out, double* cost_out, int32_t* iters_out, int32_t* status_out) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (B < 0 || n_max < 1) return fail(h, GTO_ERR_INVALID_ARG, "B must be >= 0 and n_max >= 1");
  if (B == 0) return GTO_OK;
  if (!scene_id || !qc || !goals || !n_goals || !base_pos || !Q0) return fail(h, GTO_ERR_INVALID_ARG, "null input array");
  int rc = check_scene_ids_host(h, scene_id, B);
  if (rc) return rc;
  for (int b = 0; b < B; ++b)
    if (n_goals[b] < 1 || n_goals[b] > n_max) return fail(h, GTO_ERR_INVALID_ARG, "n_goals[b] must be in [1, n_max]");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t ndof = h->rb.ndof, T = h->opts.T;
  const void *d_sid, *d_qc, *d_goals, *d_ng, *d_so, *d_base, *d_Q0;
  void *d_Q, *d_dQ, *d_cost, *d_it, *d_stat;
  if ((rc = stage_in(h, 0, scene_id, B * sizeof(int32_t), &d_sid))) return rc;
  if ((rc = stage_in(h, 1, qc, B * ndof * sizeof(double), &d_qc))) return rc;
  if ((rc = stage_in(h, 2, goals, (size_t)B * n_max * 16 * sizeof(double), &d_goals))) return rc;
  if ((rc = stage_in(h, 3, n_goals, B * sizeof(int32_t), &d_ng))) return rc;
  if ((rc = stage_in(h, 4, standoff, (size_t)B * 16 * sizeof(double), &d_so))) return rc;
  if ((rc = stage_in(h, 5, base_pos, (size_t)B * 3 * sizeof(double), &d_base))) return rc;
  if ((rc = stage_in(h, 6, Q0, B * ndof * T * sizeof(double), &d_Q0))) return rc;
  if ((rc = stage_out(h, 0, Q_out, B * ndof * T * sizeof(double), &d_Q))) return rc;
  if ((rc = stage_out(h, 1, dQ_out, B * ndof * (T - 1) * sizeof(double), &d_dQ))) return rc;
  if ((rc = stage_out(h, 2, cost_out, B * sizeof(double), &d_cost))) return rc;
  if ((rc = stage_out(h, 3, iters_out, B * sizeof(int32_t), &d_it))) return rc;
  if ((rc = stage_out(h, 4, status_out, B * sizeof(int32_t), &d_stat))) return rc;
  rc = gto_solve_batch_device(h, B, n_max, (const int32_t*)d_sid, (const double*)d_qc, (const double*)d_goals,
                              (const int32_t*)d_ng, (const double*)d_so, (const double*)d_base, (const double*)d_Q0,
                              (double*)d_Q, (double*)d_dQ, (double*)d_cost, (int32_t*)d_it, (int32_t*)d_stat, nullptr);
  if (rc) return rc;
  if ((rc = fetch_out(h, 0, Q_out, B * ndof * T * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 1, dQ_out, B * ndof * (T - 1) * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 2, cost_out, B * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 3, iters_out, B * sizeof(int32_t)))) return rc;
  if ((rc = fetch_out(h, 4, status_out, B * sizeof(int32_t)))) return rc;
  if ((rc = sync_and_finish_out(h))) return rc;
  return GTO_OK;
}

// -------------------------------------------------------------------------------------------------
int gto_eval_fk(gto_handle* h, int32_t nq, const double* q, double* frames_out) {
  if (!h || !q || !frames_out || nq < 0) return h ? fail(h, GTO_ERR_INVALID_ARG, "bad argument") : GTO_ERR_INVALID_ARG;
  if (nq == 0) return GTO_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const void* dq;
  void* dout;
  int rc;
  size_t ob = (size_t)nq * h->rb.n_frames * 16 * sizeof(double);
  if ((rc = stage_in(h, 0, q, (size_t)nq * h->rb.ndof * sizeof(double), &dq))) return rc;
  if ((rc = stage_out(h, 0, frames_out, ob, &dout))) return rc;
  {
    const size_t lds = sizeof(double) * eval_kin_lds_doubles(h->rb.n_frames, h->rb.n_links, h->rb.n_opt);
    HIPCHK(h, raise_dynamic_lds((const void*)k_eval_kin, lds));
    hipLaunchKernelGGL(k_eval_kin, dim3((nq + GTO_EVAL_TG - 1) / GTO_EVAL_TG), dim3(256), lds, h->stream, h->d_rb, nq, (const double*)dq,
                       (double*)dout, (double*)nullptr);
  }
  if ((rc = fetch_out(h, 0, frames_out, ob))) return rc;
  if ((rc = sync_and_finish_out(h))) return rc;
  return GTO_OK;
}

int gto_eval_points(gto_handle* h, int32_t scene_id, int32_t nq, const double* q, const double* base_pos, int32_t use_obs,
                    double* xyz_out, int32_t* offset_out, double* value_out, double* grad_out) {
  if (!h || !q || !base_pos || nq < 0) return h ? fail(h, GTO_ERR_INVALID_ARG, "bad argument") : GTO_ERR_INVALID_ARG;
  if (nq == 0) return GTO_OK;
  const bool want_field = offset_out || value_out || grad_out;
  if (want_field && (scene_id < 0 || (size_t)scene_id >= h->scenes.size() || !h->scenes[scene_id].valid))
    return fail(h, GTO_ERR_NO_SCENE, "unknown scene");
  HIPCHK(h, hipSetDevice(h->device));
  const int P = h->rb.n_points, L = h->rb.n_links;
  const void *dq, *dbase;
  void *dx, *doff, *dval, *dgrad;
  int rc;
  if ((rc = stage_in(h, 0, q, (size_t)nq * h->rb.ndof * sizeof(double), &dq))) return rc;
  if ((rc = stage_in(h, 1, base_pos, (size_t)nq * 3 * sizeof(double), &dbase))) return rc;
  if ((rc = ensure(h, h->vis, (size_t)nq * L * 12 * sizeof(double)))) return rc;
  if ((rc = stage_out(h, 0, xyz_out, (size_t)nq * P * 3 * sizeof(double), &dx))) return rc;
  if ((rc = stage_out(h, 1, offset_out, (size_t)nq * P * sizeof(int32_t), &doff))) return rc;
  if ((rc = stage_out(h, 2, value_out, (size_t)nq * P * sizeof(double), &dval))) return rc;
  if ((rc = stage_out(h, 3, grad_out, (size_t)nq * P * 3 * sizeof(double), &dgrad))) return rc;
  {
    const size_t lds = sizeof(double) * eval_kin_lds_doubles(h->rb.n_frames, h->rb.n_links, h->rb.n_opt);
    HIPCHK(h, raise_dynamic_lds((const void*)k_eval_kin, lds));
    hipLaunchKernelGGL(k_eval_kin, dim3((nq + GTO_EVAL_TG - 1) / GTO_EVAL_TG), dim3(256), lds, h->stream, h->d_rb, nq, (const double*)dq,
                       (double*)nullptr, (double*)h->vis.p);
  }
  hipLaunchKernelGGL(k_eval_points, dim3((P + 255) / 256, nq), dim3(256), 0, h->stream, h->d_rb, h->d_px, h->d_py, h->d_pz,
                     h->d_plink, h->d_perm, want_field ? h->d_scenes + scene_id : nullptr, nq, (const double*)h->vis.p,
                     (const double*)dbase, use_obs, (double*)dx, (int32_t*)doff, (double*)dval, (double*)dgrad);
  if ((rc = fetch_out(h, 0, xyz_out, (size_t)nq * P * 3 * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 1, offset_out, (size_t)nq * P * sizeof(int32_t)))) return rc;
  if ((rc = fetch_out(h, 2, value_out, (size_t)nq * P * sizeof(double)))) return rc;
  if ((rc = fetch_out(h, 3, grad_out, (size_t)nq * P * 3 * sizeof(double)))) return rc;
  if ((rc = sync_and_finish_out(h))) return rc;
  return GTO_OK;
}

int gto_eval_points_hessian(gto_handle* h, int32_t scene_id, int32_t nq, const double* q, const double* base_pos, int32_t use_obs,
                            double* hess_out) {
  if (!h || !q || !base_pos || !hess_out || nq < 0) return h ? fail(h, GTO_ERR_INVALID_ARG, "bad argument") : GTO_ERR_INVALID_ARG;
  if (nq == 0) return GTO_OK;
  if (scene_id < 0 || (size_t)scene_id >= h->scenes.size() || !h->scenes[scene_id].valid) return fail(h, GTO_ERR_NO_SCENE, "unknown scene");
  HIPCHK(h, hipSetDevice(h->device));
  const int P = h->rb.n_points, L = h->rb.n_links;
  const void *dq, *dbase;
  void* dh;
  int rc;
  if ((rc = stage_in(h, 0, q, (size_t)nq * h->rb.ndof * sizeof(double), &dq))) return rc;
  if ((rc = stage_in(h, 1, base_pos, (size_t)nq * 3 * sizeof(double), &dbase))) return rc;
  if ((rc = ensure(h, h->vis, (size_t)nq * L * 12 * sizeof(double)))) return rc;
  if ((rc = stage_out(h, 0, hess_out, (size_t)nq * P * 9 * sizeof(double), &dh))) return rc;
  {
    const size_t lds = sizeof(double) * eval_kin_lds_doubles(h->rb.n_frames, h->rb.n_links, h->rb.n_opt);
    HIPCHK(h, raise_dynamic_lds((const void*)k_eval_kin, lds));
    hipLaunchKernelGGL(k_eval_kin, dim3((nq + GTO_EVAL_TG - 1) / GTO_EVAL_TG), dim3(256), lds, h->stream, h->d_rb, nq, (const double*)dq,
                       (double*)nullptr, (double*)h->vis.p);
  }
  hipLaunchKernelGGL(k_eval_points_hessian, dim3((P + 255) / 256, nq), dim3(256), 0, h->stream, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_plink,
                     h->d_perm, h->d_scenes + scene_id, nq, (const double*)h->vis.p, (const double*)dbase, use_obs, (double*)dh);
  if ((rc = fetch_out(h, 0, hess_out, (size_t)nq * P * 9 * sizeof(double)))) return rc;
  return sync_and_finish_out(h);
}

// Shared by gto_eval_objective / gto_eval_obstacle_normal_eq: run init (kinematics + goal terms of Q as
// the "trial") and the obstacle kernel over all waypoints, then read the pieces back.
static int eval_common(gto_handle* h, int B, int n_max, const int32_t* scene_id, const double* goals,
                       const int32_t* n_goals, const double* standoff, const double* base_pos, const double* Q,
                       bool with_goals, std::vector<InstState>& states, std::vector<double>& blocks,
                       std::vector<double>& ssfixed) {
  int rc = check_scene_ids_host(h, scene_id, B);
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t ndof = h->rb.ndof, T = h->opts.T;
  // a neutral goal set when the caller only wants obstacle terms
  std::vector<double> dummy_goal;
  std::vector<int32_t> dummy_n;
  std::vector<double> qc(B * ndof);
  for (int b = 0; b < B; ++b)
    for (size_t i = 0; i < ndof; ++i) qc[b * ndof + i] = Q[((size_t)b * ndof + i) * T];
  if (!with_goals) {
    n_max = 1;
    dummy_goal.assign((size_t)B * 16, 0.0);
    for (int b = 0; b < B; ++b) dummy_goal[b * 16] = dummy_goal[b * 16 + 5] = dummy_goal[b * 16 + 10] = dummy_goal[b * 16 + 15] = 1.0;
    dummy_n.assign(B, 1);
    goals = dummy_goal.data();
    n_goals = dummy_n.data();
    standoff = nullptr;
  }
  const void *d_sid, *d_qc, *d_goals, *d_ng, *d_so, *d_base, *d_Q0;
  if ((rc = stage_in(h, 0, scene_id, B * sizeof(int32_t), &d_sid))) return rc;
  if ((rc = stage_in(h, 1, qc.data(), B * ndof * sizeof(double), &d_qc))) return rc;
  if ((rc = stage_in(h, 2, goals, (size_t)B * n_max * 16 * sizeof(double), &d_goals))) return rc;
  if ((rc = stage_in(h, 3, n_goals, B * sizeof(int32_t), &d_ng))) return rc;
  if ((rc = stage_in(h, 4, standoff, (size_t)B * 16 * sizeof(double), &d_so))) return rc;
  if ((rc = stage_in(h, 5, base_pos, (size_t)B * 3 * sizeof(double), &d_base))) return rc;
  if ((rc = stage_in(h, 6, Q, B * ndof * T * sizeof(double), &d_Q0))) return rc;
  if ((rc = ensure_workspace(h, B))) return rc;
  SolveParams sp = make_params(h, n_max, standoff != nullptr);
  BatchPtrs bp = make_ptrs(h, (const int32_t*)d_sid, (const double*)d_qc, (const double*)d_goals, (const int32_t*)d_ng,
                           (const double*)d_so, (const double*)d_base, (const double*)d_Q0);
  HIPCHK(h, hipMemsetAsync(bp.n_done, 0, sizeof(int32_t), h->stream));
  if (h->np == GTO_NB) hipLaunchKernelGGL(k_lm_init<GTO_NB>, dim3(B), dim3(256), 0, h->stream, h->d_rb, bp, sp, B, 1 /* raw: evaluate Q as given */);
  else hipLaunchKernelGGL(k_lm_init<16>, dim3(B), dim3(256), 0, h->stream, h->d_rb, bp, sp, B, 1);
  if ((rc = launch_obstacle(h, h->stream, bp, sp, B, 0, 4, 1, false))) return rc;
  h->last_launches = 0;
  if ((rc = launch_obstacle(h, h->stream, bp, sp, B, 2, (int)T - 2, 0, h->profiling))) return rc;
  if (h->profiling) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
    h->last_ms = ms;
  }
  states.resize(B);
  const size_t bstride = (size_t)h->np * h->np + h->np + 8;
  blocks.resize((size_t)B * T * bstride);
  ssfixed.resize((size_t)B * 4);
  HIPCHK(h, hipMemcpyAsync(states.data(), h->state.p, B * sizeof(InstState), hipMemcpyDeviceToHost, h->stream));
  // trial slot is 1 right after init (slot = 0)
  HIPCHK(h, hipMemcpyAsync(blocks.data(), (double*)h->blocks.p + (size_t)1 * B * T * bstride,
                           blocks.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(ssfixed.data(), h->ssfixed.p, ssfixed.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipGetLastError());
  return GTO_OK;
}

int gto_eval_objective(gto_handle* h, int32_t B, int32_t n_max, const int32_t* scene_id, const double* goals,
                       const int32_t* n_goals, const double* standoff, const double* base_pos, const double* Q,
                       double* f_goal, double* f_obs, double* f_vel, int32_t* goal_argmin) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (B < 0 || n_max < 1 || !scene_id || !goals || !n_goals || !base_pos || !Q) return fail(h, GTO_ERR_INVALID_ARG, "bad argument");
  if (B == 0) return GTO_OK;
  std::vector<InstState> st;
  std::vector<double> blocks, ssf;
  int rc = eval_common(h, B, n_max, scene_id, goals, n_goals, standoff, base_pos, Q, true, st, blocks, ssf);
  if (rc) return rc;
  const int T = h->opts.T;
  for (int b = 0; b < B; ++b) {
    double so = ssf[4 * b] + ssf[4 * b + 1];
    const size_t bstride = (size_t)h->np * h->np + h->np + 8, bss = (size_t)h->np * h->np + h->np;
    for (int t = 2; t < T; ++t) so += blocks[((size_t)b * T + t) * bstride + bss];
    if (f_goal) f_goal[b] = st[b].fgoal_try[0];
    if (f_obs) f_obs[b] = h->opts.w_obstacle * so;
    if (f_vel) f_vel[b] = st[b].fvel_try[0];
    if (goal_argmin) goal_argmin[b] = st[b].argmin_try[0];
  }
  return GTO_OK;
}

int gto_eval_obstacle_normal_eq(gto_handle* h, int32_t B, const int32_t* scene_id, const double* base_pos, const double* Q,
                                double* JtJ, double* Jtr, double* sumsq) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (B < 0 || !scene_id || !base_pos || !Q) return fail(h, GTO_ERR_INVALID_ARG, "bad argument");
  if (B == 0) return GTO_OK;
  std::vector<InstState> st;
  std::vector<double> blocks, ssf;
  int rc = eval_common(h, B, 1, scene_id, nullptr, nullptr, nullptr, base_pos, Q, false, st, blocks, ssf);
  if (rc) return rc;
  const int T = h->opts.T, n = h->rb.n_opt;
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      const int np = h->np;
      const double* blk = &blocks[((size_t)b * T + t) * ((size_t)np * np + np + 8)];
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j)
          if (JtJ) JtJ[(((size_t)b * T + t) * n + i) * n + j] = (t < 2) ? 0.0 : blk[np * i + j];
        if (Jtr) Jtr[((size_t)b * T + t) * n + i] = (t < 2) ? 0.0 : blk[np * np + i];
      }
      if (sumsq) sumsq[(size_t)b * T + t] = (t < 2) ? ssf[4 * b + t] : blk[np * np + np];
    }
  return GTO_OK;
}

int gto_plan_cost(gto_handle* h, int32_t scene_id, int32_t n, const double* plans, const double* base_pos, double* cost_out,
                  double* dist_out) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (n < 0 || !plans || !base_pos || !cost_out) return fail(h, GTO_ERR_INVALID_ARG, "bad argument");
  if (n == 0) return GTO_OK;
  if (scene_id < 0 || (size_t)scene_id >= h->scenes.size() || !h->scenes[scene_id].valid) return fail(h, GTO_ERR_NO_SCENE, "unknown scene");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t ndof = h->rb.ndof, T = h->opts.T;
  const void *dplans, *dbase;
  void* dpart;
  int rc;
  if ((rc = stage_in(h, 0, plans, (size_t)n * ndof * T * sizeof(double), &dplans))) return rc;
  if ((rc = stage_in(h, 1, base_pos, 3 * sizeof(double), &dbase))) return rc;
  std::vector<double> part((size_t)n * T);
  if ((rc = stage_out(h, 0, part.data(), part.size() * sizeof(double), &dpart))) return rc;
  const size_t pc_lds = sizeof(double) * plan_cost_lds_doubles(h->rb.n_frames, h->rb.n_links, h->rb.n_opt);
  if (pc_lds > 150 * 1024) return fail(h, GTO_ERR_UNSUPPORTED, "robot too large for the plan-cost kernel's LDS");
  HIPCHK(h, raise_dynamic_lds((const void*)k_plan_cost, pc_lds));
  hipLaunchKernelGGL(k_plan_cost, dim3((unsigned)((T + GTO_PLAN_TG - 1) / GTO_PLAN_TG), n), dim3(256), pc_lds, h->stream, h->d_rb, h->d_px, h->d_py, h->d_pz, h->d_plink,
                     h->d_scenes + scene_id, (int)T, (const double*)dplans, (const double*)dbase, (double*)dpart);
  if ((rc = fetch_out(h, 0, part.data(), part.size() * sizeof(double)))) return rc;
  if ((rc = sync_and_finish_out(h))) return rc;
  HIPCHK(h, hipGetLastError());
  for (int i = 0; i < n; ++i) {
    double c = 0.0, dd = 0.0;
    for (size_t t = 0; t < T; ++t) c += part[(size_t)i * T + t];  // waypoint order, like the reference loop
    for (size_t j = 0; j < ndof; ++j) {
      double v = plans[((size_t)i * ndof + j) * T] - plans[((size_t)i * ndof + j) * T + T - 1];
      dd += v * v;
    }
    cost_out[i] = c;
    if (dist_out) dist_out[i] = std::sqrt(dd);
  }
  return GTO_OK;
}


// ------------------------------------------------------------------ cost field from a depth image (row f-2)
// Device buffers of gto_depth_sdf_cost are kept between calls (the entry point has no handle to hang them on): a call
// allocates a dozen buffers, and hipMalloc / hipFree cost more than the kernels for the reference's 5 cm grids.  A buffer is
// reused for a request of at most half its size up to its size; at most 1 GiB stays cached per process.
namespace {
struct DepthPool {
  struct Item { int device; void* p; size_t cap; };
  std::mutex mu;
  std::vector<Item> items;
  size_t cached = 0;
  void* take(int device, size_t bytes, size_t* cap_out) {
    {
      std::lock_guard<std::mutex> lock(mu);
      for (size_t k = 0; k < items.size(); ++k)
        if (items[k].device == device && items[k].cap >= bytes && items[k].cap <= 2 * bytes + 4096) {
          void* p = items[k].p;
          *cap_out = items[k].cap;
          cached -= items[k].cap;
          items.erase(items.begin() + k);
          return p;
        }
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    *cap_out = bytes;
    return p;
  }
  void give(int device, void* p, size_t cap) {
    std::lock_guard<std::mutex> lock(mu);
    items.push_back({device, p, cap});
    cached += cap;
    while (cached > ((size_t)1 << 30) && !items.empty()) {  // oldest first
      (void)hipFree(items.front().p);
      cached -= items.front().cap;
      items.erase(items.begin());
    }
  }
};
DepthPool g_depth_pool;
}  // namespace

int gto_depth_sdf_cost(int device, const float* depth, int32_t H, int32_t W, const double* K, const double* Kinv,
                       const double* cam_pose, const double* cam_inv, const uint8_t* target_mask, double threshold,
                       const double* query, int64_t nq, float epsilon, float w_inside, float* sdf_out,
                       uint8_t* inside_out, float* cost_out, double* points_out, uint8_t* valid_out) {
  if (!depth || !K || !Kinv || !cam_pose || !cam_inv || H < 1 || W < 1 || nq < 0 || (nq > 0 && !query))
    return fail(nullptr, GTO_ERR_INVALID_ARG, "gto_depth_sdf_cost: null or empty input");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, GTO_ERR_NO_DEVICE, "no HIP device");
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return fail(nullptr, GTO_ERR_NO_DEVICE, "hipSetDevice failed");
  const size_t N = (size_t)H * W;
  int cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  std::vector<std::pair<void*, size_t>> bufs;
  auto dalloc = [&](size_t bytes) -> void* {
    size_t cap = 0;
    void* p = g_depth_pool.take(cur_dev, bytes ? bytes : 8, &cap);
    if (p) bufs.emplace_back(p, cap);
    return p;
  };
  auto cleanup = [&]() {
    (void)hipDeviceSynchronize();  // nothing of this call may still be using them when the next call takes them
    for (auto& b : bufs) g_depth_pool.give(cur_dev, b.first, b.second);
  };
#define DCHK(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      cleanup();                                                                                       \
      return fail(nullptr, GTO_ERR_HIP, std::string("gto_depth_sdf_cost: ") + hipGetErrorString(e_));  \
    }                                                                                                  \
  } while (0)
  float* d_depth = (float*)dalloc(N * sizeof(float));
  double* d_mats = (double*)dalloc((9 + 9 + 16 + 16) * sizeof(double));
  uint8_t* d_mask = target_mask ? (uint8_t*)dalloc(N) : nullptr;
  double* d_p = (double*)dalloc(3 * N * sizeof(double));
  uint8_t* d_valid = (uint8_t*)dalloc(N);
  double* d_q = (double*)dalloc((size_t)nq * 3 * sizeof(double));
  float* d_sdf = (float*)dalloc((size_t)nq * sizeof(float));
  float* d_cost = (float*)dalloc((size_t)nq * sizeof(float));
  uint8_t* d_in = (uint8_t*)dalloc((size_t)nq);
  if (!d_depth || !d_mats || (target_mask && !d_mask) || !d_p || !d_valid || !d_q || !d_sdf || !d_cost || !d_in) {
    cleanup();
    return fail(nullptr, GTO_ERR_ALLOC, "gto_depth_sdf_cost: device allocation failed");
  }
  double mats[50];
  std::memcpy(mats, K, 9 * sizeof(double));
  std::memcpy(mats + 9, Kinv, 9 * sizeof(double));
  std::memcpy(mats + 18, cam_pose, 16 * sizeof(double));
  std::memcpy(mats + 34, cam_inv, 16 * sizeof(double));
  DCHK(hipMemcpy(d_depth, depth, N * sizeof(float), hipMemcpyHostToDevice));
  DCHK(hipMemcpy(d_mats, mats, sizeof mats, hipMemcpyHostToDevice));
  if (target_mask) DCHK(hipMemcpy(d_mask, target_mask, N, hipMemcpyHostToDevice));
  if (nq) DCHK(hipMemcpy(d_q, query, (size_t)nq * 3 * sizeof(double), hipMemcpyHostToDevice));
  double *d_px = d_p, *d_py = d_p + N, *d_pz = d_p + 2 * N;
  hipLaunchKernelGGL(k_depth_backproject, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, d_depth, H, W, d_mats + 9,
                     d_mats + 18, d_mask, threshold, d_px, d_py, d_pz, d_valid);
  const char* brute_env = getenv("GTO_DEPTH_BRUTE");
  const bool brute = brute_env && atoi(brute_env) != 0;  // the exhaustive search (reference construction; the two are compared in a test)
  if (nq && brute) {
    hipLaunchKernelGGL(k_depth_sdf, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, d_px, d_py, d_pz, (int)N, d_depth, H, W,
                       d_mats, d_mats + 34, d_q, (long)nq, epsilon, w_inside, d_sdf, d_in, d_cost);
  } else if (nq) {
    // bounding-box hierarchy over 8 x 4 pixel tiles of the depth image (k_depth_sdf_bvh): same distances, bit for bit
    const int tx = (W + GTO_BVH_TILE_W - 1) / GTO_BVH_TILE_W, ty = (H + GTO_BVH_TILE_H - 1) / GTO_BVH_TILE_H;
    int P = 1;
    while (P < tx || P < ty) P <<= 1;
    if (P > 1024) {  // more tiles per side than k_bvh_up's single workgroup builds: the exhaustive search serves such images
      hipLaunchKernelGGL(k_depth_sdf, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, d_px, d_py, d_pz, (int)N, d_depth, H, W,
                         d_mats, d_mats + 34, d_q, (long)nq, epsilon, w_inside, d_sdf, d_in, d_cost);
    } else {
    double* d_boxes = (double*)dalloc((size_t)(2 * P * P) * 6 * sizeof(double));
    if (!d_boxes) {
      cleanup();
      return fail(nullptr, GTO_ERR_ALLOC, "gto_depth_sdf_cost: device allocation failed");
    }
    hipLaunchKernelGGL(k_bvh_leaves, dim3((unsigned)((P * P + 255) / 256)), dim3(256), 0, 0, d_px, d_py, d_pz, H, W, P, d_boxes);
    if (P > 1) hipLaunchKernelGGL(k_bvh_up, dim3(1), dim3(1024), 0, 0, P, d_boxes);
    // queries in Morton order of their position (coherent waves): 30-bit keys, hipCUB radix sort of (key, index)
    if (nq >= ((int64_t)1 << 31)) {
      cleanup();
      return fail(nullptr, GTO_ERR_UNSUPPORTED, "gto_depth_sdf_cost: more than 2^31 queries");
    }
    unsigned* d_keys = (unsigned*)dalloc((size_t)nq * 4 * sizeof(unsigned));  // keys in / out, indices in / out
    if (!d_keys) {
      cleanup();
      return fail(nullptr, GTO_ERR_ALLOC, "gto_depth_sdf_cost: device allocation failed");
    }
    unsigned long long* d_stats = nullptr;
    if (getenv("GTO_DEPTH_STATS")) {
      d_stats = (unsigned long long*)dalloc(3 * sizeof(unsigned long long));
      if (d_stats) DCHK(hipMemset(d_stats, 0, 3 * sizeof(unsigned long long)));
    }
    unsigned *d_keys2 = d_keys + nq, *d_idx = d_keys + 2 * nq, *d_idx2 = d_keys + 3 * nq;
    hipLaunchKernelGGL(k_query_keys, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, d_q, (long)nq, d_boxes, d_keys, d_idx);
    size_t tmp_bytes = 0;
    DCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, (int)nq, 0, 30, (hipStream_t)0));
    void* d_tmp = dalloc(tmp_bytes);
    if (!d_tmp) {
      cleanup();
      return fail(nullptr, GTO_ERR_ALLOC, "gto_depth_sdf_cost: device allocation failed");
    }
    DCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, (int)nq, 0, 30, (hipStream_t)0));
    hipLaunchKernelGGL(k_depth_sdf_bvh, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, d_px, d_py, d_pz, d_boxes, P, d_idx2, d_depth, H, W,
                       d_mats, d_mats + 34, d_q, (long)nq, epsilon, w_inside, d_sdf, d_in, d_cost, d_stats, 0);
    if (d_stats) {
      unsigned long long st[3];
      DCHK(hipMemcpy(st, d_stats, sizeof st, hipMemcpyDeviceToHost));
      fprintf(stderr, "[gto] depth field search: %lld queries, nodes popped per query %.1f, leaves per query %.1f, loop iterations per wave %.1f\n",
              (long long)nq, (double)st[0] / nq, (double)st[1] / nq, (double)st[2] / ((nq + 63) / 64));
    }
    }
  }
  DCHK(hipGetLastError());
  DCHK(hipDeviceSynchronize());
  if (sdf_out && nq) DCHK(hipMemcpy(sdf_out, d_sdf, (size_t)nq * sizeof(float), hipMemcpyDeviceToHost));
  if (cost_out && nq) DCHK(hipMemcpy(cost_out, d_cost, (size_t)nq * sizeof(float), hipMemcpyDeviceToHost));
  if (inside_out && nq) DCHK(hipMemcpy(inside_out, d_in, (size_t)nq, hipMemcpyDeviceToHost));
  if (valid_out) DCHK(hipMemcpy(valid_out, d_valid, N, hipMemcpyDeviceToHost));
  if (points_out) {
    std::vector<double> soa(3 * N);
    DCHK(hipMemcpy(soa.data(), d_p, 3 * N * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; ++i)
      for (int r = 0; r < 3; ++r) points_out[3 * i + r] = soa[(size_t)r * N + i];
  }
#undef DCHK
  cleanup();
  return GTO_OK;
}


// numpy.arange(start, stop, step) for doubles, value for value: the length is ceil((stop - start) / step), the fill is
// a[i] = start + i * ((start + step) - start) (numpy's DOUBLE_fill takes the increment from the first two elements)
static std::vector<double> np_arange(double start, double stop, double step) {
  const double len = std::ceil((stop - start) / step);
  const long n = len > 0 ? (long)len : 0;
  std::vector<double> a((size_t)n);
  const double delta = (start + step) - start;
  for (long i = 0; i < n; ++i) a[i] = i == 0 ? start : (i == 1 ? start + step : start + (double)i * delta);
  return a;
}

/* include/gto_solver.h: the per-object perception steps of examples/pybullet_gto_planning.py:176-190 in one call, with
 * nothing but the grid geometry coming back to the host.  The depth image goes up once; the cloud of all pixels and the
 * cloud without the target's pixels are back-projected from it; the grid is the bounding box of the first cloud plus
 * `margin` at `grid_res` (gto/gto_models.py:155-171, numpy.arange's values); both cost fields are searched with ONE
 * ordering of the voxel centres (one key pass, one radix sort) against the two tile hierarchies, and installed as scene
 * `scene_id` with their voxel records and distance fields, device to device. */
int gto_scene_from_depth(gto_handle* h, int32_t scene_id, const float* depth, int32_t H, int32_t W, const double* K,
                         const double* Kinv, const double* cam_pose, const double* cam_inv, const uint8_t* target_mask,
                         const float* depth_obstacle, double threshold, double grid_res, double margin, float epsilon,
                         float w_inside, int32_t* shape_out, double* origin_out, double* bounds_out) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (!depth || !K || !Kinv || !cam_pose || !cam_inv || H < 1 || W < 1 || !(grid_res > 0) || !(margin >= 0))
    return fail(h, GTO_ERR_INVALID_ARG, "gto_scene_from_depth: null or empty input");
  HIPCHK(h, hipSetDevice(h->device));
  const bool stats = getenv("GTO_DEPTH_STATS") != nullptr;
  auto t_now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_[8] = {t_now(), 0, 0, 0, 0, 0, 0, 0};
  const size_t N = (size_t)H * W;
  const int tx = (W + GTO_BVH_TILE_W - 1) / GTO_BVH_TILE_W, ty = (H + GTO_BVH_TILE_H - 1) / GTO_BVH_TILE_H;
  int P = 1;
  while (P < tx || P < ty) P <<= 1;
  if (P > 1024) return fail(h, GTO_ERR_UNSUPPORTED, "gto_scene_from_depth: image larger than 8192 x 4096 pixels");
  std::vector<std::pair<void*, size_t>> bufs;
  auto dalloc = [&](size_t bytes) -> void* {
    size_t cap = 0;
    void* p = g_depth_pool.take(h->device, bytes ? bytes : 8, &cap);
    if (p) bufs.emplace_back(p, cap);
    return p;
  };
  auto cleanup = [&]() {
    (void)hipDeviceSynchronize();
    for (auto& b : bufs) g_depth_pool.give(h->device, b.first, b.second);
  };
#define DCHK(expr)                                                                                   \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      cleanup();                                                                                     \
      return fail(h, GTO_ERR_HIP, std::string("gto_scene_from_depth: ") + hipGetErrorString(e_));    \
    }                                                                                                \
  } while (0)
#define DNULL(p)                                                                                     \
  do {                                                                                               \
    if (!(p)) {                                                                                      \
      cleanup();                                                                                     \
      return fail(h, GTO_ERR_ALLOC, "gto_scene_from_depth: device allocation failed");               \
    }                                                                                                \
  } while (0)
  // the second cloud: the obstacle image (the driver's depth_obstacle, examples/pybullet_gto_planning.py:187-189: the target's
  // pixels pushed to the threshold) without the masked pixels; its visibility test reads the obstacle image
  const bool two = target_mask != nullptr || depth_obstacle != nullptr;
  float* d_depth = (float*)dalloc(N * sizeof(float));
  float* d_depth_o = depth_obstacle ? (float*)dalloc(N * sizeof(float)) : d_depth;
  double* d_mats = (double*)dalloc((9 + 9 + 16 + 16 + 8) * sizeof(double));
  uint8_t* d_mask = target_mask ? (uint8_t*)dalloc(N) : nullptr;
  double* d_pa = (double*)dalloc(3 * N * sizeof(double));
  double* d_po = two ? (double*)dalloc(3 * N * sizeof(double)) : d_pa;
  uint8_t* d_valid = (uint8_t*)dalloc(N);
  double* d_boxa = (double*)dalloc((size_t)(2 * P * P) * 6 * sizeof(double));
  double* d_boxo = two ? (double*)dalloc((size_t)(2 * P * P) * 6 * sizeof(double)) : d_boxa;
  DNULL(d_depth); DNULL(d_depth_o); DNULL(d_mats); DNULL(d_pa); DNULL(d_po); DNULL(d_valid); DNULL(d_boxa); DNULL(d_boxo);
  if (target_mask) DNULL(d_mask);
  double mats[50];
  std::memcpy(mats, K, 9 * sizeof(double));
  std::memcpy(mats + 9, Kinv, 9 * sizeof(double));
  std::memcpy(mats + 18, cam_pose, 16 * sizeof(double));
  std::memcpy(mats + 34, cam_inv, 16 * sizeof(double));
  DCHK(hipMemcpy(d_depth, depth, N * sizeof(float), hipMemcpyHostToDevice));
  if (depth_obstacle) DCHK(hipMemcpy(d_depth_o, depth_obstacle, N * sizeof(float), hipMemcpyHostToDevice));
  DCHK(hipMemcpy(d_mats, mats, sizeof mats, hipMemcpyHostToDevice));
  if (target_mask) DCHK(hipMemcpy(d_mask, target_mask, N, hipMemcpyHostToDevice));
  t_[1] = t_now();
  const unsigned nbN = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(k_depth_backproject, dim3(nbN), dim3(256), 0, 0, d_depth, H, W, d_mats + 9, d_mats + 18, (const uint8_t*)nullptr, threshold,
                     d_pa, d_pa + N, d_pa + 2 * N, d_valid);
  // the hierarchy of the first cloud: its root box is the bounding box of the valid points (gto/gto_models.py:155-157)
  hipLaunchKernelGGL(k_bvh_leaves, dim3((unsigned)((P * P + 255) / 256)), dim3(256), 0, 0, d_pa, d_pa + N, d_pa + 2 * N, H, W, P, d_boxa);
  if (P > 1) hipLaunchKernelGGL(k_bvh_up, dim3(1), dim3(1024), 0, 0, P, d_boxa);
  if (two) {
    hipLaunchKernelGGL(k_depth_backproject, dim3(nbN), dim3(256), 0, 0, d_depth_o, H, W, d_mats + 9, d_mats + 18, (const uint8_t*)d_mask, threshold,
                       d_po, d_po + N, d_po + 2 * N, d_valid);
    hipLaunchKernelGGL(k_bvh_leaves, dim3((unsigned)((P * P + 255) / 256)), dim3(256), 0, 0, d_po, d_po + N, d_po + 2 * N, H, W, P, d_boxo);
    if (P > 1) hipLaunchKernelGGL(k_bvh_up, dim3(1), dim3(1024), 0, 0, P, d_boxo);
  }
  double root[6];
  DCHK(hipMemcpy(root, d_boxa, sizeof root, hipMemcpyDeviceToHost));  // (synchronises with the null stream)
  t_[2] = t_now();
  if (!(root[0] <= root[3]) || !std::isfinite(root[0]) || !std::isfinite(root[3])) {
    cleanup();
    return fail(h, GTO_ERR_INVALID_ARG, "gto_scene_from_depth: no valid pixel in the depth image");
  }
  std::vector<double> ax[3];
  int32_t shape[3];
  double origin[3];
  size_t nq = 1;
  for (int a = 0; a < 3; ++a) {
    ax[a] = np_arange(root[a] - margin, root[3 + a] + margin, grid_res);
    shape[a] = (int32_t)ax[a].size();
    origin[a] = root[a] - margin;
    nq *= ax[a].size();
  }
  if (nq == 0 || nq >= ((size_t)1 << 31)) {
    cleanup();
    return fail(h, GTO_ERR_UNSUPPORTED, "gto_scene_from_depth: empty grid or more than 2^31 voxels");
  }
  std::vector<double> axes(ax[0]);
  axes.insert(axes.end(), ax[1].begin(), ax[1].end());
  axes.insert(axes.end(), ax[2].begin(), ax[2].end());
  double* d_axes = (double*)dalloc(axes.size() * sizeof(double));
  double* d_q = (double*)dalloc(nq * 3 * sizeof(double));
  float* d_costa = (float*)dalloc(nq * sizeof(float));
  float* d_costo = two ? (float*)dalloc(nq * sizeof(float)) : d_costa;
  float* d_sdf = (float*)dalloc(nq * sizeof(float));
  uint8_t* d_in = (uint8_t*)dalloc(nq);
  unsigned* d_keys = (unsigned*)dalloc(nq * 4 * sizeof(unsigned));
  DNULL(d_axes); DNULL(d_q); DNULL(d_costa); DNULL(d_costo); DNULL(d_sdf); DNULL(d_in); DNULL(d_keys);
  DCHK(hipMemcpy(d_axes, axes.data(), axes.size() * sizeof(double), hipMemcpyHostToDevice));
  const unsigned nbq = (unsigned)((nq + 255) / 256);
  hipLaunchKernelGGL(k_grid_queries, dim3(nbq), dim3(256), 0, 0, d_axes, shape[0], shape[1], shape[2], d_q);
  unsigned *d_keys2 = d_keys + nq, *d_idx = d_keys + 2 * nq, *d_idx2 = d_keys + 3 * nq;
  hipLaunchKernelGGL(k_query_keys, dim3(nbq), dim3(256), 0, 0, d_q, (long)nq, d_boxa, d_keys, d_idx);
  size_t tmp_bytes = 0;
  DCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, (int)nq, 0, 30, (hipStream_t)0));
  void* d_tmp = dalloc(tmp_bytes);
  DNULL(d_tmp);
  DCHK(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, (int)nq, 0, 30, (hipStream_t)0));
  unsigned long long* d_stats = nullptr;
  if (stats) {
    d_stats = (unsigned long long*)dalloc(3 * sizeof(unsigned long long));
    if (d_stats) DCHK(hipMemset(d_stats, 0, 3 * sizeof(unsigned long long)));
    DCHK(hipDeviceSynchronize());
  }
  t_[3] = t_now();
  // the two searches are independent and each is bound by its slowest packets (the voxels deep behind the surfaces): side
  // by side on two streams of their own, behind everything the null stream has done so far
  static std::mutex s_mu;
  static std::vector<std::pair<int, std::pair<hipStream_t, hipStream_t>>> s_streams;  // per device, kept for the process
  hipStream_t sa = nullptr, sb = nullptr;
  {
    std::lock_guard<std::mutex> lock(s_mu);
    for (auto& e : s_streams)
      if (e.first == h->device) sa = e.second.first, sb = e.second.second;
    if (!sa) {
      DCHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
      DCHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
      s_streams.push_back({h->device, {sa, sb}});
    }
  }
  DCHK(hipStreamSynchronize(0));
  uint8_t* d_in2 = two ? (uint8_t*)dalloc(nq) : d_in;
  DNULL(d_in2);
  hipLaunchKernelGGL(k_depth_sdf_bvh, dim3(nbq), dim3(256), 0, sa, d_pa, d_pa + N, d_pa + 2 * N, d_boxa, P, d_idx2, d_depth, H, W, d_mats, d_mats + 34,
                     d_q, (long)nq, epsilon, w_inside, (float*)nullptr, d_in, d_costa, d_stats, 1);
  if (two)
    hipLaunchKernelGGL(k_depth_sdf_bvh, dim3(nbq), dim3(256), 0, sb, d_po, d_po + N, d_po + 2 * N, d_boxo, P, d_idx2, d_depth_o, H, W, d_mats, d_mats + 34,
                       d_q, (long)nq, epsilon, w_inside, (float*)nullptr, d_in2, d_costo, (unsigned long long*)nullptr, 1);
  DCHK(hipGetLastError());
  DCHK(hipDeviceSynchronize());
  t_[4] = t_now();
  if (d_stats) {
    unsigned long long stv[3];
    DCHK(hipMemcpy(stv, d_stats, sizeof stv, hipMemcpyDeviceToHost));
    fprintf(stderr, "[gto] depth field search (first cloud): %zu queries, nodes popped per wave %.1f, leaves per wave %.1f\n", nq, (double)stv[2] / ((nq + 63) / 64), (double)stv[1] / 64 / ((nq + 63) / 64));
  }
  const int rc = set_scene_impl(h, scene_id, d_costa, two ? d_costo : nullptr, shape, origin, grid_res, false, hipMemcpyDeviceToDevice);
  t_[5] = t_now();
#undef DCHK
#undef DNULL
  cleanup();
  t_[6] = t_now();
  if (stats)
    fprintf(stderr, "[gto] scene from depth (%d x %d image, %zu voxels), ms: alloc + upload %.3f | back-projection, hierarchies, bounds %.3f | queries, keys, sort %.3f | "
                    "two searches %.3f | records + distance fields %.3f | release %.3f | total %.3f\n",
            H, W, nq, t_[1] - t_[0], t_[2] - t_[1], t_[3] - t_[2], t_[4] - t_[3], t_[5] - t_[4], t_[6] - t_[5], t_[6] - t_[0]);
  if (rc) return rc;
  if (shape_out) std::memcpy(shape_out, shape, sizeof shape);
  if (origin_out) std::memcpy(origin_out, origin, sizeof origin);
  if (bounds_out) std::memcpy(bounds_out, root, sizeof root);
  return GTO_OK;
}

/* The two cost fields of a resident scene, device to host (float32 [nx ny nz] each; either pointer may be null). */
int gto_get_scene_fields(gto_handle* h, int32_t scene_id, float* c_all_out, float* c_obs_out) {
  if (!h) return GTO_ERR_INVALID_ARG;
  if (scene_id < 0 || (size_t)scene_id >= h->scenes.size() || !h->scenes[scene_id].valid)
    return fail(h, GTO_ERR_NO_SCENE, "gto_get_scene_fields: the scene was never set");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const SceneDev& s = h->scenes[scene_id];
  const size_t nvox = (size_t)s.nx * s.ny * s.nz;
  if (c_all_out) HIPCHK(h, hipMemcpy(c_all_out, s.c_all, nvox * sizeof(float), hipMemcpyDeviceToHost));
  if (c_obs_out) HIPCHK(h, hipMemcpy(c_obs_out, s.c_obs, nvox * sizeof(float), hipMemcpyDeviceToHost));
  return GTO_OK;
}

}  // extern "C"
