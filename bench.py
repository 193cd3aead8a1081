#!/usr/bin/env python3
"""bench.py — throughput of the GTO inner solve on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: solving B independent (scene, goal-grasp)
trajectory problems of BASELINE.json configs[1] (Panda 7-DoF, T=50 waypoints, ~5k surface points,
128^3 float32 cost field, 64 candidate goal grasps of one scene) with the batched Gauss-Newton/LM
solver behind the C ABI (gto_solve_batch_device: inputs already resident in HBM).
N GPUs = N ranks (`python bench.py --gpus N` spawns them itself when it is not already running under
torch.distributed.run), each solving its own scene x 64 grasps (weak scaling, no data-path collective);
with N > 1 the scene-sharded workload of BASELINE configs[3] (many scenes x 8 grasps, grouped by scene,
results all_gathered over RCCL) is measured next to it ("scene_sharded").
The K timed steps run through grasptrajopt_amd.parallel.BatchPipeline with --pipeline D lanes per GPU
(one solver handle + stream + host thread each, every lane its own grasp sets): a single batch of 64 is a
latency-bound chain of launches that leaves most CUs idle, and consecutive batches are independent.  The
strictly serial rate (one batch per call) is reported next to it in "pipeline".

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel: bytes of the surface points it ACTUALLY gathered (device counter) over the
                HIP-event time of its launches; the reference's per-point work that the broad phase proves
                to be exact zeros is reported as alg_bytes_skipped_frac, not priced
  cpu_baseline  the CPU oracle (same algorithm, FP64, OpenMP over instances) on a bounded sample, with the
                same quality block as the GPU result
Exit code 3 if the quality gate fails (joint limits, plan cost against the seed).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # same guide: half the FP32 vector rate


def plan_calls(n, lanes, merge):
    """Split n steps (batches) into solver calls of at most `merge` batches, a multiple of `lanes` calls when n allows,
    sizes within one of each other: every lane gets the same amount of work and exactly n steps are run."""
    if n <= 0:
        return []
    L = max(-(-n // merge), min(n, lanes))
    if L % lanes and n >= lanes * (L // lanes + 1):
        L = lanes * (L // lanes + 1)
    return [n // L + (1 if i < n % L else 0) for i in range(L)]


LANE_CORES, RANK_CORES = 0.27, 0.5  # measured: cores a lane's host thread takes, cores of a rank's main thread


def lanes_that_fit(asked, world, quota):
    """Lanes per rank that fit the CPUs the cgroup may use: world x (RANK_CORES + lanes x LANE_CORES) <= quota (None: no limit).
    Returns (lanes, the decision in words)."""
    if world <= 1 or quota is None:
        return asked, f"{asked} lanes per rank as asked for"
    fit = int((quota / world - RANK_CORES) / LANE_CORES + 1e-9)
    if fit < asked:
        lanes = max(1, fit)
        return lanes, (f"{lanes} lanes per rank: cgroup CPU quota {quota} / {world} ranks = {quota / world:.2f} cores per rank < "
                       f"{RANK_CORES} + {asked} lanes x {LANE_CORES} cores (measured per lane)")
    return asked, (f"{asked} lanes per rank as asked for: {world} ranks x ({RANK_CORES} + {asked} x {LANE_CORES} measured cores per lane) = "
                   f"{world * (RANK_CORES + asked * LANE_CORES):.1f} <= cgroup CPU quota {quota}")


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: launch N ranks on this node (one per GPU) the way the
    driver does and hand their output through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n}: spawning {n} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr)
    return subprocess.call(cmd)


def quality_block(desc, cfg, h, sid, Q, RT, Q0, iters, status, max_iter, in_collision_c=0.01):
    """SURVEY.md 8d quality gate on solved trajectories Q (n, ndof, T): joint limits, goal pose against the IK thresholds
    of examples/pybullet_gto_planning.py:262, compute_plan_cost against the seed, and the evaluator's collision
    statistic (examples/pybullet_evaluate_plans.py:219-239: a waypoint with more than 5 body points inside an obstacle;
    inside <=> cost > epsilon / 2 on the synthetic field) on the first 64 plans."""
    oi = desc.opt_index
    viol = float(np.maximum(desc.lower[oi][None, :, None] - Q[:, oi], Q[:, oi] - desc.upper[oi][None, :, None]).max())
    fe = desc.frame_index(cfg["link_ee"])
    Tf = h.eval_fk(Q[:, :, -1])[:, fe]
    err_pos = np.linalg.norm(Tf[:, :3, 3] - RT[:, :3, 3], axis=1)
    cosang = (np.einsum("bij,bij->b", Tf[:, :3, :3], RT[:, :3, :3]) - 1.0) / 2.0
    err_rot = np.degrees(np.arccos(np.clip(cosang, -1, 1)))
    seed_cost, _ = h.plan_cost(sid, Q0, [0, 0, 0])
    sol_cost, _ = h.plan_cost(sid, Q, [0, 0, 0])
    ns = min(64, Q.shape[0])
    moving = desc.link_is_moving()[desc.point_link]
    hit = 0
    for i in range(ns):
        _, _, val, _ = h.eval_points(sid, Q[i].T, [0.0, 0.0, 0.0], use_obs=True)
        hit += int((((val > in_collision_c) & moving[None, :]).sum(axis=1) > 5).any())
    ok = (err_pos < 0.01) & (err_rot < 5)
    miss = ~ok
    return {"max_joint_limit_violation": viol, "goal_err_pos_max_m": round(float(err_pos.max()), 5),
            "goal_err_rot_max_deg": round(float(err_rot.max()), 3), "goal_ok_frac": round(float(ok.mean()), 3),
            "plan_cost_le_seed_frac": round(float((sol_cost <= seed_cost + 1e-12).mean()), 3),
            "plans_in_collision_frac": round(hit / ns, 3), "plans_checked_for_collision": ns,
            # how the instances that miss the 1 cm / 5 deg goal thresholds ended: at the iteration cap or converged
            "goal_miss_ended_at_max_iter_frac": round(float((iters[miss] >= max_iter).mean()), 3) if miss.any() else None,
            "goal_miss_converged_frac": round(float((status[miss] == 0).mean()), 3) if miss.any() else None}


def lib_sha16():
    """sha256 (first 16 hex digits) over the sources the HIP library is built from, in the order of __graft_entry__.HIP_DEPS:
    ties a bench line (and every file under profiles/ that carries it) to the code it measured."""
    import hashlib
    import __graft_entry__ as g
    hsh = hashlib.sha256()
    for d in g.HIP_DEPS:
        with open(d if os.path.isabs(d) else os.path.join(g.CSRC, d), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


LINE_LIMIT = 6000  # bytes of the final stdout line (the driver's record keeps an 8 KB tail; round 5's 21.6 KB line did not parse)


def _pick(d, keys):
    return None if d is None else {k: d[k] for k in keys if k in d}


def _dig(d, *keys):
    for k in keys:
        d = d.get(k) if isinstance(d, dict) else None
    return d


def compact_line(out):
    """The ONE JSON line the driver parses: the contract's keys plus the figures a reader needs to recompute `roofline.frac`
    and to see the quality gate, under LINE_LIMIT bytes.  Everything else (per-variant tables, next_rows, prose) is in
    bench_detail.json beside this script and on the `BENCH_DETAIL` line printed before this one."""
    rf = out.get("roofline") or {}
    dom = rf.get("dominant_by_time") or {}
    alg = rf.get("alg_bytes_per_launch")
    cb = out.get("cpu_baseline")
    q = out.get("quality") or {}
    ha = out.get("host_api")
    pl = out.get("pipeline") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit")}
    line["metric_definition"] = "value: inputs/outputs resident in HBM (gto_solve_batch_device); host_api: host pointers, H2D/D2H timed (SURVEY 8d literal)"
    line.update({k: out.get(k) for k in ("n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")})
    line["config"] = _pick(out.get("config"), ("workload", "batch_per_gpu", "T", "surface_points", "grid", "max_iter", "steps_per_call", "lanes", "parallelism"))
    if line["config"] and isinstance(line["config"].get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:240]
    line["roofline"] = dict(_pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_unit", "points_gathered_per_launch",
                                       "alg_bytes_per_launch", "alg_bytes_skipped_frac", "avg_launch_us", "launches", "traffic_calibration")) or {},
                            traffic_over_alg_bytes=(round(rf["traffic"] / alg, 2) if rf.get("traffic") and alg else None),
                            timed_regime_frac=(rf.get("timed_regime") or {}).get("frac"),
                            dominant_by_time=_pick(dom, ("kernel", "share_of_solve_loop_kernel_time", "avg_launch_us", "workgroups_per_launch", "frac",
                                                         "valu_issue_frac", "waves_per_simd", "critical_path_over_launch")))
    if line["roofline"]["dominant_by_time"] is not None and "frac" not in line["roofline"]["dominant_by_time"]:
        line["roofline"]["dominant_by_time"]["bound"] = "latency"
    line["cpu_baseline"] = None if cb is None else dict(
        _pick(cb, ("value", "unit", "cores", "kind", "iters_per_s", "max_abs_dQ_vs_gpu", "iters_equal_gpu")),
        sample=cb.get("sample", "")[:120], single_thread=(cb.get("single_thread") or {}).get("value"),
        goal_ok_frac=(cb.get("quality") or {}).get("goal_ok_frac"))
    line["timed_regions"] = _pick(out.get("timed_regions"), ("repeats", "what", "value_min", "value_max", "spread_rel"))
    line["host_api"] = _pick(ha, ("trajectories_per_s", "ms_per_step", "vs_device_resident"))
    line["pipeline"] = _pick(pl, ("lanes", "steps_per_call", "slots_per_lane", "serial_ms_per_step", "host_cpu_cores_busy", "lane_rates",
                                  "lane_results_reproducible_alone", "merged_equals_single_batch_solves"))
    line.update({k: out.get(k) for k in ("sqp_iters_per_s", "iters_mean", "iters_max", "status_counts")})
    line["quality"] = dict(_pick(q, ("gate", "goal_ok_frac", "max_joint_limit_violation", "objective_le_seed_frac", "plan_cost_le_seed_frac",
                                     "plans_in_collision_frac", "goal_err_pos_max_m", "goal_err_rot_max_deg")) or {},
                           f_excess_rel_max_converged=(q.get("stopping_tolerance_check") or {}).get("f_excess_rel_max_converged"),
                           goal_sets_of_8=_pick(q.get("goal_sets_of_8"), ("goal_ok_frac", "iters_mean", "plans_in_collision_frac")),
                           reference_shaped=_pick(q.get("reference_shaped"), ("gate", "goal_ok_frac", "goal_ok_ge_0.95", "goal_pos_ok_frac", "goal_ok_frac_cpu_port", "plans_in_collision_frac",
                                                                              "iters_mean", "chord_rad_5_50_95", "chord_rad_stored_5_50_95", "instances", "goals_per_instance")))
    oc = out.get("other_configs")
    if oc is not None:
        line["other_configs"] = {}
        for k, v in oc.items():
            if "error" in v:
                line["other_configs"][k] = {"error": v["error"]}
                continue
            r2, ck = v.get("roofline") or {}, v.get("oracle_check") or {}
            line["other_configs"][k] = {"trajectories_per_s": v.get("trajectories_per_s"), "ms_per_step": v.get("ms_per_step"),
                                        "kernel": r2.get("kernel"), "frac": r2.get("frac"), "achieved": r2.get("achieved"), "avg_launch_us": r2.get("avg_launch_us"),
                                        "points_gathered_per_launch": r2.get("points_gathered_per_launch"), "traffic": r2.get("traffic"),
                                        "dominant_by_time": _pick(r2.get("dominant_by_time"), ("kernel", "share_of_solve_loop_kernel_time", "avg_launch_us")),
                                        "iters_mean": v.get("iters_mean"), "gate": v.get("gate"),
                                        "oracle_check_ok": bool(ck.get("iters_equal") and ck.get("status_equal") and ck.get("max_abs_dQ_vs_oracle", 1.0) < 1e-6),
                                        "max_abs_dQ_vs_oracle": ck.get("max_abs_dQ_vs_oracle")}
    else:
        line["other_configs"] = None
    nr = out.get("next_rows")
    if nr is not None:  # one figure per SURVEY 8(f) row; the rest is in the detail file
        g_ = lambda *ks: _dig(nr, *ks)
        line["next_rows"] = {"f1_ik_per_s_with_collision": g_("f1_ik", "with_collision_term", "ik_per_s"),
                             "f2_depth_field_128_ms": g_("f2_depth_cost_field", "grid_128", "ms"),
                             "f3_plan_scores_per_s": g_("f3_plan_scores", "plan_scores_per_s"),
                             "f4_base_sets_per_s_w0.01": g_("f4_base_placement", "effort_weight_0.01", "sets_per_s"),
                             "plan_goalset_call_ms": g_("a1_plan_goalset_call", "ms_median"),
                             "per_object_pipeline_ms": g_("per_object_pipeline", "device_resident", "ms", "per_object"),
                             "oracle_checks_ok": bool(g_("f1_ik", "with_collision_term", "check", "iters_equal") and g_("f2_depth_cost_field", "grid_128", "check", "cost_bit_identical_to_oracle")
                                                      and g_("f4_base_placement", "effort_weight_0.01", "check", "iters_equal") and g_("a1_plan_goalset_call", "check", "iters_equal"))}
    ss = out.get("scene_sharded")
    line["scene_sharded"] = _pick(ss, ("instances", "trajectories_per_s", "scenes_per_rank", "fields_resident_gb_this_rank", "all_instances_returned",
                                       "own_shard_round_trip_exact", "iters_mean", "max_joint_limit_violation", "work_queue"))
    col = out.get("collective")
    line["collective"] = _pick(col, ("backend", "is_rccl", "world_size", "nccl_version", "distinct_devices", "cgroup_cpu_quota_cores", "lanes_per_rank",
                                     "lane_decision", "host_cpu_cores_busy_all_ranks"))
    line["oracle_check"] = _pick(out.get("oracle_check"), ("instances", "iters_equal", "status_equal", "max_abs_dQ_vs_oracle"))
    line["reference_published"] = "0.098 trajectories/s (BASELINE.md section 1, unknown CPU)"
    line["detail"] = "bench_detail.json (also the BENCH_DETAIL line above)"
    line["lib_sha16"] = out.get("lib_sha16")
    s_ = json.dumps(line)
    if len(s_) >= LINE_LIMIT:  # never again a line the driver cannot keep: drop the optional objects, largest first
        for k in ("next_rows", "pipeline", "scene_sharded", "collective", "other_configs", "oracle_check"):
            line[k] = None
            s_ = json.dumps(line)
            if len(s_) < LINE_LIMIT:
                break
    return line


def emit_lines(out):
    """Detail to bench_detail.json and to an earlier stdout line, the compact line LAST."""
    try:
        with open(os.path.join(ROOT, "bench_detail.json"), "w") as fh:
            json.dump(out, fh)
    except OSError as e:
        print(f"[bench] bench_detail.json not written: {e}", file=sys.stderr)
    print("BENCH_DETAIL " + json.dumps(out))
    sys.stdout.flush()
    print(json.dumps(compact_line(out)))
    sys.stdout.flush()


def main(argv=None, emit=True):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="goal grasps (instances) per GPU per step")
    ap.add_argument("--pipeline", type=int, default=4, help="lanes per GPU: solver handles, each with its stream and host thread")
    ap.add_argument("--lanes", type=int, default=0, help="lanes INSIDE a solver call (gto_set_lanes), each on a torch stream of its own; 0: the library's default")
    ap.add_argument("--merge", type=int, default=32, help="steps (batches) a lane hands to the solver in one call; the solver keeps "
                    "GTO_SLOTS (512) of their instances in flight and refills slots as instances finish")
    ap.add_argument("--max-iter", type=int, default=100, help="iteration cap (reference IPOPT cap: 100)")
    ap.add_argument("--robot", default="panda_5k")
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--T", type=int, default=50, help="waypoints (reference: 50); the standoff waypoint stays a fifth of them from the end")
    ap.add_argument("--shelf", action="store_true", help="shelf scene (boards and walls around the objects) instead of a table top")
    ap.add_argument("--mode", choices=["rounds"], default="rounds", help="solver mode (include/gto_solver.h GTO_MODE_*; the single-launch mode was removed in round 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the next_rows object (IK, depth field, seed scoring, base placement, one plan_goalset call)")
    ap.add_argument("--merged-launches-only", action="store_true",
                    help="skip the one-batch-per-launch passes (serial latency, host API): every launch of the run then has the "
                         "timed region's size, which is what the per-launch PMC averages of tools/pmc_pass.sh need")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="target duration of the CPU-baseline sample")
    ap.add_argument("--scenes-per-gpu", type=int, default=0,
                    help="scene-sharded leg (N > 1, or --scene-sharded): scenes per GPU, 8 grasps each; 0 = 256 with --gpus 8 "
                         "(BASELINE configs[3]: 2048 scenes over 8 GPUs, 37.9 GB of fields resident per GPU), else 32")
    ap.add_argument("--scene-sharded", action="store_true", help="run the scene-sharded leg also on one GPU")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the N > 1 run: nccl = RCCL (one GPU per rank); gloo = host-side collectives, "
                         "for dry runs of the multi-rank code on a box with fewer GPUs than ranks")
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 code path (process group, barriers, all_reduce of the timings, the scene-sharded leg's all_gather, the collective object) with however many ranks there are, even one: the RCCL branch on a one-GPU box")
    ap.add_argument("--same-device", action="store_true",
                    help="every rank uses device 0 (with --backend gloo: the N > 1 code path on ONE GPU; the rates then say nothing)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps, barrier + synchronize on both sides) is run this many times "
                         "back to back; ms_per_step / value are the MEDIAN region, min and max are reported beside it")
    ap.add_argument("--light", action="store_true",
                    help="a second workload inside another bench line (other_configs): timed regions, roofline, invariants and an "
                         "oracle spot check only")
    ap.add_argument("--oracle-check", type=int, default=0, help="instances of the first batch solved again by the CPU oracle (iterations equal, max |dQ|)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs object (BASELINE configs[2] and [4] at reduced step counts)")
    args = ap.parse_args(argv)
    child = argv is not None  # a workload measured inside another bench line: no process group, nothing printed

    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not child:
        raise SystemExit(spawn_ranks(args.gpus))

    # the HIP runtime multiplexes streams onto this many hardware queues (default 4); the pipeline lanes
    # must not share one, or their launches serialise
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    rank = 0 if child else int(os.environ.get("RANK", "0"))
    world = 1 if child else int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1 or (args.force_dist and not child and "WORLD_SIZE" in os.environ)  # a process group exists
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the GTO solve path has no CPU fallback")
    if args.same_device:
        if args.backend == "nccl":
            raise SystemExit("--same-device needs --backend gloo: RCCL refuses two ranks on one device")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the tensors of the (few, tiny) collectives live
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import __graft_entry__ as g
    if "GTO_HIP_LIB" not in os.environ:  # (an A/B run names its own library)
        if world > 1 and rank != 0:  # one node: rank 0 builds, the others wait for it
            dist.barrier()
        g.build()  # mtime-checked against every source: a stale library is never what gets measured
        if world > 1 and rank == 0:
            dist.barrier()
    from grasptrajopt_amd import _capi, synthetic as syn
    from grasptrajopt_amd.parallel import BatchPipeline, shard_by_scene, shard_range, solve_local_shard
    from grasptrajopt_amd.robot_desc import load_builtin

    fetch = args.robot.startswith("fetch")  # BASELINE configs[2]: --robot fetch --batch 256 (shelf-height table)
    cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", f"{args.robot.split('_')[0]}_cfg.json")))
    desc = load_builtin(args.robot)
    opts = _capi.default_opts()
    opts.max_iter = args.max_iter
    opts.T = args.T
    opts.standoff_offset = -max(2, args.T // 5)  # reference: T = 50, offset -10 (gto/gto_planner.py:22,25)
    T, ndof, B = opts.T, desc.ndof, args.batch
    mobile = args.robot.endswith("_mobile")      # BASELINE configs[4]: --robot fetch_mobile --T 80 --grid 256 --shelf
    default_pose = np.concatenate([np.zeros(ndof - len(cfg["default_pose"])), np.array(cfg["default_pose"], dtype=np.float64)])
    D, M = max(1, args.pipeline), max(1, args.merge)

    def cpu_quota():
        try:  # cgroup v2: "max 100000" or "<quota> <period>"
            q_, p_ = open("/sys/fs/cgroup/cpu.max").read().split()
            return None if q_ == "max" else round(int(q_) / int(p_), 2)
        except Exception:
            return None
    # every lane is a host thread that sleep-polls two pinned words: LANE_CORES of a core each, measured (host_cpu_cores_busy
    # 1.05-1.06 for four lanes on the driver's call, 0.66 on the default run: pipeline.host_cpu_cores_busy of every line);
    # N ranks x D lanes plus RANK_CORES for a rank's main thread must fit the CPUs this cgroup may use, or the lanes' naps
    # turn into scheduling delays on every rank.  (Round 5 cut to quota // ranks lanes -- one core per lane, four times
    # what a lane takes: 2 lanes per rank at quota 16 / 8 ranks, where 8 x (0.5 + 4 x 0.27) = 12.6 cores fit.)
    quota_ = cpu_quota()
    D, lane_decision = lanes_that_fit(D, world, quota_)
    slots = int(os.environ.get("GTO_SLOTS", "512"))  # instances a solver call keeps in flight (gto_api.hip)
    mode = _capi.SolverHandle.MODE_ROUNDS
    kernel_name = "k_obstacle_gram"

    # this rank's shard of the global problem list: scene = global rank id, 64 grasps each
    lo, hi = shard_range(world * B, rank, world)
    assert hi - lo == B
    scene_seed = lo // B
    res = 2.24 / args.grid  # covers the 2.24 m reach box (0.0175 m at 128^3)
    origin = (-0.3, -1.12, 0.0) if fetch else (-0.4, -1.12, -0.4)
    table_z = 0.45 if fetch else -0.03
    if args.shelf:
        table_z = 0.75
    if mobile:  # the base roams +-1 m: a 4.48 m box around it
        res, origin = 4.48 / args.grid, (-1.6, -2.24, -0.2)
    make_scene = lambda seed: syn.make_scene(seed, n=args.grid, res=res, origin=origin, table_z=table_z, shelf=args.shelf)
    sc = make_scene(scene_seed)

    # D pipeline lanes: each one solver handle bound to ONE stream of its own, with its own M batches (M consecutive
    # steps, grasp sets of its own) in HBM and its own outputs; a call solves m <= M of them at once
    class Lane:
        def __init__(self, first=None):
            # streams of the greatest priority: the runtime gives them hardware queues of their own, in the order of their first
            # use, so the lanes sit behind different dispatcher pipes whatever other streams the process has created (a lane
            # whose queue shares a pipe with another's runs its evaluation launches 1.5x slower: DESIGN.md section 6)
            self.stream = torch.cuda.Stream(dev, priority=int(os.environ.get("GTO_BENCH_STREAM_PRIO", "-1")))
            self.h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=local_rank, n_gripper_points=100)
            self.h.set_mode(mode)
            self.h.set_stream(self.stream.cuda_stream)
            if args.lanes > 0:
                self.lane_streams = [self.stream] + [torch.cuda.Stream(dev, priority=int(os.environ.get("GTO_BENCH_STREAM_PRIO", "-1"))) for _ in range(args.lanes - 1)]
                self.h.set_lanes(args.lanes, int(os.environ.get("GTO_LANE_MIN", "256")), int(os.environ.get("GTO_ADOPT", "48")))
                self.h.set_lane_streams([s_.cuda_stream for s_ in self.lane_streams])
            if first is None:
                self.h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
            else:
                self.h.share_scene(0, first.h)  # one copy of the scene in HBM for all lanes

        def upload(self, qc, RT, S, base, Q0):
            n = qc.shape[0]  # M * B instances, batch m in rows [m B, (m+1) B)
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
            self.host = (np.zeros(n, np.int32), qc, RT.reshape(n, 1, 16), np.ones(n, np.int32), S, base, Q0)
            self.inp = [torch.zeros(n, dtype=torch.int32, device=dev), t(qc, torch.float64), t(RT.reshape(n, 1, 16), torch.float64),
                        torch.ones(n, dtype=torch.int32, device=dev), t(S, torch.float64), t(base, torch.float64), t(Q0, torch.float64)]
            self.d_Q = torch.empty((n, ndof, T), dtype=torch.float64, device=dev)
            self.d_dQ = torch.empty((n, ndof, T - 1), dtype=torch.float64, device=dev)
            self.d_cost = torch.empty(n, dtype=torch.float64, device=dev)
            self.d_it = torch.empty(n, dtype=torch.int32, device=dev)
            self.d_st = torch.empty(n, dtype=torch.int32, device=dev)
            self.bufs = self.inp + [self.d_Q, self.d_dQ, self.d_cost, self.d_it, self.d_st]

        def step(self, m=1, first=0):
            """One call over batches first .. first+m-1 of this lane (device-resident entry point)."""
            ptrs = [x.data_ptr() + first * B * x.stride(0) * x.element_size() for x in self.bufs]
            self.h.solve_batch_device(m * B, 1, *ptrs, self.stream.cuda_stream)

        def host_step(self, m=1, first=0):
            """The same call through the host-pointer entry point (H2D / D2H of the per-instance data included); the result
            arrays are the lane's own, kept from call to call like its device buffers."""
            if getattr(self, "host_out", None) is None:
                n = self.host[1].shape[0]
                self.host_out = (np.empty((n, ndof, T)), np.empty((n, ndof, T - 1)), np.empty(n), np.empty(n, np.int32), np.empty(n, np.int32))
            return self.h.solve_batch(*[a[first * B:(first + m) * B] for a in self.host], out=tuple(a[first * B:(first + m) * B] for a in self.host_out))

    lanes = [Lane()]
    lanes += [Lane(lanes[0]) for _ in range(D - 1)]
    h = lanes[0].h
    # goal grasps: collision-free configurations w.r.t. the obstacle field (target object removed);
    # links that no optimised joint moves (the base) are ignored
    moving = desc.link_is_moving()[desc.point_link]

    def goal_collision_cost(q):
        _, _, val, _ = h.eval_points(0, q, [0.0, 0.0, 0.0], use_obs=True)
        return (val * moving[None, :]).sum(axis=1)

    zlim = (0.55, 1.2) if fetch else (0.08, 0.7)
    if args.shelf:
        zlim = (table_z + 0.07, table_z + 0.33)
    NB = M * B
    qc = np.tile(default_pose, (NB, 1))
    S = np.tile(syn.standoff_pose(-0.1, cfg["axis_standoff"]).reshape(1, 16), (NB, 1))
    base = np.zeros((NB, 3))
    lane_data = []
    for d_, ln in enumerate(lanes):  # every lane its own M grasp sets of the scene: nothing is solved twice at the same time
        sets = [syn.make_goals(desc, h.eval_fk, cfg["link_ee"], B, seed=scene_seed + 1009 * m + 500009 * d_,
                               collision_cost=goal_collision_cost, zlim=zlim) for m in range(M)]
        RT_, qg_ = np.concatenate([x[0] for x in sets]), np.concatenate([x[1] for x in sets])
        Q0_ = np.stack([syn.make_seed(qc[b], qg_[b], T, desc.param_index) for b in range(NB)])
        if args.shelf:  # shelf scenes are planned with interpolate=False (examples/pybullet_gto_planning.py:98-109,
            # gto/gto_planner.py:216-219): hold qc, jump to the IK solution at the standoff waypoint
            hold = np.repeat(qc[:, :, None], T, axis=2)
            hold[:, :, T + opts.standoff_offset:] = Q0_[:, :, -1:]
            Q0_ = hold
        ln.upload(qc, RT_, S, base, Q0_)
        lane_data.append((RT_, Q0_))
    RT, Q0 = lane_data[0]
    torch.cuda.synchronize(dev)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    launch_plan = lambda n: plan_calls(n, D, M)

    def run_steps(pipe, n, method="step"):
        futs = [pipe.submit(method, m) for m in launch_plan(n)]
        return [f.result() for f in futs]

    pipe = BatchPipeline(lanes)
    run_steps(pipe, args.warmup * D * M)  # every lane sees >= W warmup launches
    barrier()
    R = max(1, args.repeats)

    def timed_regions(method):
        """R timed regions back to back, each exactly K steps (batches) with up to D x M of them in flight, each bracketed
        by barrier + synchronize; every rank takes the same decisions (the times are reduced over the ranks below)."""
        out, cpu = [], []
        for _ in range(R):
            barrier()
            t0, c0 = time.perf_counter(), time.process_time()
            run_steps(pipe, args.steps, method)
            barrier()
            out.append(time.perf_counter() - t0)
            cpu.append(time.process_time() - c0)
        return np.array(out), np.array(cpu)

    # ---- timed regions, device-resident entry point (inputs in HBM when the region starts): `value`
    el_all, cpu_all = timed_regions("step")
    # ---- SURVEY.md 8d's literal metric: the same K steps through the host-pointer entry point on the same lanes
    host_all = None
    if args.light:
        args.merged_launches_only = True
    if not args.merged_launches_only:
        # warm-up of the host-pointer path: its staging buffers are allocated by the first call, and one of the first few
        # calls after that takes 5-10 ms longer (measured, once per process; not in the library's own code path)
        for _ in range(max(1, min(args.warmup, 3))):
            run_steps(pipe, min(args.steps, D * M), "host_step")
        host_all, _ = timed_regions("host_step")
    pipe.close()
    # ---- the same K steps on ONE and on TWO of the lanes (N = 1 lines only): the per-GPU rate a rank would run at if the
    # host budget of an 8-rank job left it fewer lanes than it asks for (DESIGN.md section 9's prediction rests on these)
    lane_rates = None
    if world == 1 and not child and not args.light and not args.merged_launches_only and D > 1:
        lane_rates = {}
        for nl in sorted({1, 2, D} - {D}):
            sub = BatchPipeline(lanes[:nl])
            D_keep, D = D, nl  # launch_plan splits the K steps over the lanes in use
            try:
                run_steps(sub, min(args.steps, nl * M))
                ts_ = []
                for _ in range(3):
                    barrier()
                    t0_ = time.perf_counter()
                    run_steps(sub, args.steps)
                    barrier()
                    ts_.append(time.perf_counter() - t0_)
            finally:
                D = D_keep
                sub.close()
            lane_rates[str(nl)] = round(B * args.steps / float(np.median(ts_)), 1)
        lane_rates[str(D)] = round(B * args.steps / float(np.median(el_all)), 1)
    ln0 = lanes[0]
    # a lane's results do not depend on what the other lanes do: lane 0 solves lane 1's batches again, alone on the GPU
    lanes_reproducible = None
    if D > 1 and not args.light:
        m1 = max(launch_plan(args.steps))
        ref_Q, ref_it = lanes[1].d_Q[:m1 * B].clone(), lanes[1].d_it[:m1 * B].clone()
        keep = [x.clone() for x in ln0.inp]
        for dst, src in zip(ln0.inp, lanes[1].inp):
            dst.copy_(src)
        torch.cuda.synchronize(dev)  # the copies run on torch's stream, the solve on the lane's
        ln0.step(m1)
        barrier()
        lanes_reproducible = bool(torch.equal(ln0.d_Q[:m1 * B], ref_Q)) and bool(torch.equal(ln0.d_it[:m1 * B], ref_it))
        for dst, src in zip(ln0.inp, keep):
            dst.copy_(src)
        torch.cuda.synchronize(dev)
    ln0.step(max(launch_plan(args.steps)))
    barrier()
    merged_Q, merged_it = ln0.d_Q.clone(), ln0.d_it.clone()

    # ---- the same K steps strictly one after the other, one batch per call, on one lane: per-batch
    # latency (and a check that a batch solved alone equals the batch solved inside a merged call)
    barrier()
    ts = time.perf_counter()
    merged_equals_single = None
    if not args.merged_launches_only:
        for k in range(args.steps):
            ln0.step(1, k % M)
        barrier()
        covered = min(args.steps, M, max(launch_plan(args.steps))) * B
        merged_equals_single = bool(torch.equal(ln0.d_Q[:covered], merged_Q[:covered])) and bool(torch.equal(ln0.d_it[:covered], merged_it[:covered]))
    serial_elapsed = time.perf_counter() - ts
    # ---- the calls of the timed region once more, alone on the GPU, with HIP events around every launch
    # of the dominant kernel (on its launch stream) for the roofline: a launch's duration is then the
    # kernel's own and matches the rocprofv3 summary; the kernel counts the surface points it gathers
    h.set_profiling(True)
    kern_ms, kern_launches, prof_steps, gathered, tested = 0.0, 0, 0, 0, 0
    plan = launch_plan(args.steps)
    variants = {}  # kernel variant -> [ms, launches, workgroups, points gathered] over the calls of the region
    for m in plan:
        ln0.step(m)
        ms, nl = h.last_kernel_time()
        pts, tst = h.last_kernel_work()
        kern_ms += ms
        kern_launches += nl
        gathered += pts
        tested += tst
        prof_steps += m
        if args.mode == "rounds":
            for name_, row in h.last_kernel_profile().items():
                acc_ = variants.setdefault(name_, [0.0, 0, 0, 0])
                for i_ in range(4):
                    acc_[i_] += row[i_]
    barrier()
    h.set_profiling(False)
    ln0.step(M)  # full outputs again for the quality gate
    barrier()
    d_it, d_st, d_cost, d_Q = ln0.d_it, ln0.d_st, ln0.d_cost, ln0.d_Q

    # one batch per call through the host-pointer entry point (latency of the drop-in call)
    host_rate = None
    if rank == 0 and not args.merged_launches_only:
        hs = max(2, min(5, args.steps))
        ln0.host_step(1)
        th = time.perf_counter()
        for _ in range(hs):
            ln0.host_step(1)
        host_rate = B * hs / (time.perf_counter() - th)

    iters = d_it.cpu().numpy().astype(np.int64)
    status = d_st.cpu().numpy()
    cost = d_cost.cpu().numpy()
    Qsol = d_Q.cpu().numpy()
    el = torch.tensor(np.concatenate([el_all, host_all if host_all is not None else np.zeros(0)]), dtype=torch.float64, device=comm_dev)
    # iterations done inside the timed region: a call of m steps solves a lane's batches 0 .. m-1 (lane 0's counts stand
    # for the other lanes' grasp sets of the same scene)
    per_batch_it = iters.reshape(M, B).sum(axis=1)
    it_timed = float(sum(per_batch_it[:m].sum() for m in plan))
    it_sum = torch.tensor([it_timed], dtype=torch.float64, device=comm_dev)
    if multi:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(it_sum, op=dist.ReduceOp.SUM)
    el_np = el.cpu().numpy()  # per region: the slowest rank's time
    el_all = el_np[:R]
    host_all = el_np[R:] if host_all is not None else None
    elapsed = float(np.median(el_all))
    host_cpu = float(np.median(cpu_all))
    host_pipe_rate = None if host_all is None else B * args.steps / float(np.median(host_all))
    total_traj = world * B * args.steps
    value = total_traj / elapsed
    iters_per_s = float(it_sum.item()) / elapsed

    # ---- BASELINE configs[3]: many scenes x 8 grasps, instances grouped by scene onto ranks, solved through the
    # host-pointer API, results all_gathered (RCCL); every instance of a call reads a different field
    scene_sharded = None
    if multi or args.scene_sharded:
        SG = 8
        SP = args.scenes_per_gpu if args.scenes_per_gpu > 0 else (256 if world == 8 else 32)
        n_sc = SP * world
        nI = n_sc * SG
        sid_all = np.repeat(np.arange(n_sc, dtype=np.int32), SG)
        owner = shard_by_scene(sid_all, world)
        mine = np.nonzero(owner == rank)[0]            # this rank's instances (global indices), grouped by scene
        mine_sc = np.unique(sid_all[mine])
        hs_ = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=local_rank, n_gripper_points=100)
        hs_.set_mode(mode)
        free0 = torch.cuda.mem_get_info(dev)[0]
        RTl, qgl = np.empty((len(mine), 4, 4)), np.empty((len(mine), ndof))  # arguments of THIS rank's instances only
        for s_ in mine_sc:
            scs = make_scene(100 + int(s_))
            hs_.set_scene(int(s_), scs.c_all, scs.c_obs, scs.shape, scs.origin, scs.res)

            def cc(q, s_=int(s_)):
                _, _, val, _ = hs_.eval_points(s_, q, [0.0, 0.0, 0.0], use_obs=True)
                return (val * moving[None, :]).sum(axis=1)
            rows = np.nonzero(sid_all[mine] == s_)[0]
            RTl[rows], qgl[rows] = syn.make_goals(desc, hs_.eval_fk, cfg["link_ee"], SG, seed=7000 + int(s_), collision_cost=cc, zlim=zlim)
        resident_gb = (free0 - torch.cuda.mem_get_info(dev)[0]) / 1e9
        qcl = np.tile(default_pose, (len(mine), 1))
        Q0l = np.stack([syn.make_seed(qcl[i], qgl[i], T, desc.param_index) for i in range(len(mine))])
        Sl = np.tile(syn.standoff_pose(-0.1, cfg["axis_standoff"]).reshape(1, 16), (len(mine), 1))
        sargs = (mine, sid_all[mine], qcl, RTl.reshape(len(mine), 1, 16), np.ones(len(mine), np.int32), Sl, np.zeros((len(mine), 3)), Q0l)
        solve_local_shard(hs_.solve_batch, *sargs, B=nI, rank=rank, world=world, assignment=owner)  # warm-up
        barrier()
        tss = time.perf_counter()
        gstats = {}
        idx, Qa, _, ca, ia, sa = solve_local_shard(hs_.solve_batch, *sargs, B=nI, rank=rank, world=world, assignment=owner, stats=gstats)
        barrier()
        els = torch.tensor([time.perf_counter() - tss], dtype=torch.float64, device=comm_dev)
        if multi:
            dist.all_reduce(els, op=dist.ReduceOp.MAX)
        oi_ = desc.opt_index
        scene_sharded = {"workload": f"BASELINE configs[3]: {n_sc} scenes x {SG} grasps, grouped by scene onto {world} rank(s) "
                                     f"({SP} scenes per rank), host-pointer API, every rank builds and solves its own instances only, results "
                                     + (f"all_gathered over {'RCCL' if args.backend == 'nccl' else 'gloo'}" if multi else "kept on the one rank"),
                         "instances": int(nI), "trajectories_per_s": round(nI / float(els.item()), 1),
                         "scenes_per_rank": int(SP), "fields_resident_gb_this_rank": round(resident_gb, 2),
                         "all_instances_returned": bool(len(idx) == nI and np.array_equal(idx, np.arange(nI))),
                         "own_shard_round_trip_exact": bool(np.array_equal(Qa[mine], solve_local_shard(hs_.solve_batch, *sargs, B=nI, rank=rank, world=world, assignment=owner, gather=False)[1])),
                         "iters_mean": round(float(ia.mean()), 2), "status_counts": {str(k): int((sa == k).sum()) for k in np.unique(sa)},
                         "max_joint_limit_violation": float(np.maximum(desc.lower[oi_][None, :, None] - Qa[:, oi_], Qa[:, oi_] - desc.upper[oi_][None, :, None]).max())}
        hs_.close()

    # ---- N > 1: what the collectives ran on, so that the first multi-GPU record shows "RCCL saw N ranks on N distinct
    # devices" and whether N x (lanes + polling host threads) fit the host: gathered from every rank
    collective = None
    if multi:
        try:
            bus = torch.cuda.get_device_properties(dev).pci_bus_id
        except Exception:
            bus = None
        mine_ = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(), "pci_bus_id": bus,
                 "device_name": torch.cuda.get_device_name(dev), "host_cpu_cores_busy": round(float(np.median(cpu_all)) / float(np.median(el_all)), 2),
                 "pid": os.getpid(), "all_gather": dict(gstats)}  # the scene-sharded leg always runs with N > 1
        rows_ = [None] * world
        dist.all_gather_object(rows_, mine_)
        nccl_v = None
        if args.backend == "nccl":
            try:
                nccl_v = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                nccl_v = None
        collective = {"backend": dist.get_backend(), "is_rccl": args.backend == "nccl", "world_size": dist.get_world_size(),
                      "nccl_version": nccl_v, "ranks": rows_,
                      "distinct_devices": len({(r_["device_index"], r_["pci_bus_id"]) for r_ in rows_}),
                      "cgroup_cpu_quota_cores": cpu_quota(), "host_hardware_threads": os.cpu_count(),
                      "lanes_per_rank": D, "lane_decision": lane_decision,
                      "host_cpu_cores_busy_all_ranks": round(sum(r_["host_cpu_cores_busy"] for r_ in rows_), 2),
                      "data_path_collectives": "none inside the solve; one all_gather pair (float64 trajectories, int32 iterations/status) of the "
                                               "scene-sharded leg's results, two all_reduce of timings"}

    rc = 0
    if rank == 0:
        quality = quality_block(desc, cfg, h, 0, Qsol, RT, Q0, iters, status, args.max_iter)
        quality["f_mean"] = round(float(cost.mean()), 5)
        # objective split of the instances that miss the goal thresholds: the velocity term outweighs the goal term there
        fgo, _, fve, _ = h.eval_objective(0, RT.reshape(NB, 1, 16), 1, S[0].reshape(4, 4), [0.0, 0.0, 0.0], Qsol)
        quality["f_goal_mean"], quality["f_vel_mean"] = round(float(fgo.mean()), 5), round(float(fve.mean()), 5)
        # solver invariant: the objective at the returned trajectory is never above the (clipped, pinned) seed's
        seedc = Q0.copy()
        oi_g = desc.opt_index
        seedc[:, oi_g] = np.clip(seedc[:, oi_g], desc.lower[oi_g][None, :, None], desc.upper[oi_g][None, :, None])
        seedc[:, :, :2] = qc[:, :, None]
        sg_, so_, sv_, _ = h.eval_objective(0, RT.reshape(NB, 1, 16), 1, S[0].reshape(4, 4), [0.0, 0.0, 0.0], seedc)
        quality["objective_le_seed_frac"] = round(float((cost <= (sg_ + so_ + sv_) * (1 + 1e-12)).mean()), 4)
        light_gate = None
        if args.light:  # a second workload inside another bench line: the invariants only
            light_gate = quality["max_joint_limit_violation"] <= 1e-8 and quality["objective_le_seed_frac"] == 1.0
        gate_ok = bool(light_gate)
        if not args.light:
            # what the shipped stopping tolerance (tol_rel_f) leaves on the table: the first batch again with a tolerance five
            # orders tighter and three times the iteration cap, objective against objective
            tol0, it0 = opts.tol_rel_f, opts.max_iter
            h.set_opts(tol_rel_f=tol0 * 1e-5, max_iter=3 * it0)
            _, _, f_tight, it_tight, _ = h.solve_batch(0, qc[:B], RT[:B].reshape(B, 1, 16), 1, S[:B], base[:B], Q0[:B])
            h.set_opts(tol_rel_f=tol0, max_iter=it0)
            gap = (cost[:B] - f_tight) / np.maximum(f_tight, 1e-300)
            quality["stopping_tolerance_check"] = {"tol_rel_f": tol0, "tight_tol_rel_f": tol0 * 1e-5, "tight_max_iter": 3 * it0,
                                                   "f_excess_rel_max": float(gap.max()), "f_excess_rel_mean": float(gap.mean()),
                                                   "f_excess_rel_max_converged": float(gap[status[:B] == 0].max()) if (status[:B] == 0).any() else None,
                                                   "iters_mean": round(float(iters[:B].mean()), 2), "iters_mean_tight": round(float(it_tight.mean()), 2)}
            # gate: joint limits and the objective invariant always; compute_plan_cost against the seed's where the seed is a
            # collision-scored trajectory (interpolated seeds; the hold-and-jump seeds of shelf scenes cost nothing by construction)
            gate_ok = quality["max_joint_limit_violation"] <= 1e-8 and quality["objective_le_seed_frac"] == 1.0
            # ... and the stopping tolerance may not cost a converged instance more than 1e-4 of its objective
            gate_ok = gate_ok and (quality["stopping_tolerance_check"]["f_excess_rel_max_converged"] or 0.0) <= 1e-4
            if not args.shelf:
                gate_ok = gate_ok and quality["plan_cost_le_seed_frac"] >= 0.95
            # ---- what the north star's gradient choice buys: the first batch again with GTO_GRAD_ZERO, the reference-faithful
            # obstacle gradient (CasADi differentiates neither floor() nor the parametric gather: SURVEY.md Appendix B-1), same
            # statistics on the same 64 instances next to the shipped GTO_GRAD_CENTRAL_DIFF
            gm0 = opts.grad_mode
            same = (0, qc[:B], RT[:B].reshape(B, 1, 16), 1, S[:B], base[:B], Q0[:B])
            by_grad = {}
            for name_, gm in (("central_diff", _capi.GTO_GRAD_CENTRAL_DIFF), ("zero", _capi.GTO_GRAD_ZERO)):
                h.set_opts(grad_mode=gm)
                Qg_, _, fg_, itg_, stg_ = h.solve_batch(*same)
                qb_ = quality_block(desc, cfg, h, 0, Qg_, RT[:B], Q0[:B], itg_.astype(np.int64), stg_, args.max_iter)
                by_grad[name_] = {k_: qb_[k_] for k_ in ("goal_ok_frac", "goal_err_pos_max_m", "goal_err_rot_max_deg", "plans_in_collision_frac",
                                                          "plan_cost_le_seed_frac", "max_joint_limit_violation")}
                by_grad[name_].update({"f_mean": round(float(fg_.mean()), 5), "iters_mean": round(float(itg_.mean()), 2)})
            h.set_opts(grad_mode=gm0)
            quality["by_obstacle_gradient"] = dict(by_grad, what="first batch (64 instances) solved with each grad_mode; `zero` is what reaches IPOPT "
                                                                 "in the reference (value only), `central_diff` is shipped (gto/sdf_callback.py:90-114 numerics)")
            gate_ok = gate_ok and by_grad["central_diff"]["plans_in_collision_frac"] <= by_grad["zero"]["plans_in_collision_frac"]
            # ---- goal sets of eight, what plan_goalset is called with (examples/pybullet_gto_planning.py:291): instance b gets
            # eight grasps of the lane's list, its seed is the least colliding / shortest of the eight interpolated plans
            # (gto/gto_planner.py:197-213), the goal error is taken against the goal the solver ends at (arg-min of the set)
            G8 = 8
            if NB >= B * G8 and not args.shelf:
                RT8 = RT[:B * G8].reshape(B, G8, 4, 4)
                seeds8 = Q0[:B * G8].reshape(B, G8, ndof, T)
                pick = np.zeros(B, dtype=np.int64)
                for b_ in range(B):
                    pc_, pd_ = h.plan_cost(0, seeds8[b_], [0.0, 0.0, 0.0])
                    pick[b_] = int(np.lexsort((pd_, pc_))[0])
                Q08 = seeds8[np.arange(B), pick]
                args8 = (0, qc[:B], RT8.reshape(B, G8, 16), G8, S[:B], base[:B], Q08)
                h.solve_batch(*args8)
                t8 = time.perf_counter()
                Q8, _, f8, it8, st8 = h.solve_batch(*args8)
                t8 = time.perf_counter() - t8
                _, _, _, am8 = h.eval_objective(0, RT8.reshape(B, G8, 16), G8, S[0].reshape(4, 4), [0.0, 0.0, 0.0], Q8)
                qb8 = quality_block(desc, cfg, h, 0, Q8, RT8[np.arange(B), am8], Q08, it8.astype(np.int64), st8, args.max_iter)
                quality["goal_sets_of_8"] = dict({k_: qb8[k_] for k_ in ("goal_ok_frac", "goal_err_pos_max_m", "goal_err_rot_max_deg", "plans_in_collision_frac",
                                                                           "plan_cost_le_seed_frac", "max_joint_limit_violation")},
                                                 instances=B, goals_per_instance=G8, iters_mean=round(float(it8.mean()), 2), iters_max=int(it8.max()),
                                                 f_mean=round(float(f8.mean()), 5), ms_one_call_host_api=round(1e3 * t8, 3),
                                                 trajectories_per_s_one_call=round(B / t8, 1), goals_reached_distinct=int(len(np.unique(am8))),
                                                 what="64 instances x goal sets of 8 grasps (plan_goalset's call shape), one solver call through the host-pointer API")
                gate_ok = gate_ok and qb8["max_joint_limit_violation"] <= 1e-8
            # ---- a reference-shaped workload: goal sets of eight as plan_goalset is called
            # (examples/pybullet_gto_planning.py:291), the grasps of a set close to each other like the grasps of one object, the
            # set's distance from qc drawn from the stored plans' chord distribution (1.6-3.5 rad: plan_statistics.npz `chord`;
            # the throughput workload's goals are 2.9-5.0 rad away); seed = least colliding / shortest of the eight interpolated
            # plans (gto/gto_planner.py:197-213); goal error against the goal the solver ends at
            if not args.shelf and not mobile and args.robot.split("_")[0] in syn.STORED_CHORD_RAD:
                Gr = 8
                RTr, Qgr = syn.make_goal_sets_reference_shaped(desc, h.eval_fk, cfg["link_ee"], default_pose, B, Gr, seed=scene_seed,
                                                               chord_rad=syn.STORED_CHORD_RAD[args.robot.split("_")[0]], zlim=zlim,
                                                               collision_cost=goal_collision_cost)
                seedsr = np.stack([[syn.make_seed(default_pose, Qgr[b_, g_], T, desc.param_index) for g_ in range(Gr)] for b_ in range(B)])
                pickr = np.zeros(B, dtype=np.int64)
                for b_ in range(B):
                    pc_, pd_ = h.plan_cost(0, seedsr[b_], [0.0, 0.0, 0.0])
                    pickr[b_] = int(np.lexsort((pd_, pc_))[0])
                Q0r = seedsr[np.arange(B), pickr]
                argsr = (0, qc[:B], RTr.reshape(B, Gr, 16), Gr, S[:B], base[:B], Q0r)
                h.solve_batch(*argsr)
                tr_ = time.perf_counter()
                Qr, _, fr, itr, str_ = h.solve_batch(*argsr)
                tr_ = time.perf_counter() - tr_
                _, _, _, amr = h.eval_objective(0, RTr.reshape(B, Gr, 16), Gr, S[0].reshape(4, 4), [0.0, 0.0, 0.0], Qr)
                qbr = quality_block(desc, cfg, h, 0, Qr, RTr[np.arange(B), amr], Q0r, itr.astype(np.int64), str_, args.max_iter)
                chord_ = np.linalg.norm(Qgr[:, :, desc.opt_index] - default_pose[desc.opt_index], axis=2)
                fe_ = desc.frame_index(cfg["link_ee"])
                Tf_ = h.eval_fk(Qr[:, :, -1])[:, fe_]
                pos_ok = float((np.linalg.norm(Tf_[:, :3, 3] - RTr[np.arange(B), amr][:, :3, 3], axis=1) < 0.01).mean())
                quality["reference_shaped"] = dict(
                    {k_: qbr[k_] for k_ in ("goal_ok_frac", "goal_err_pos_max_m", "goal_err_rot_max_deg", "plans_in_collision_frac", "plan_cost_le_seed_frac",
                                            "max_joint_limit_violation", "goal_miss_converged_frac")},
                    instances=B, goals_per_instance=Gr, goal_pos_ok_frac=round(pos_ok, 3), iters_mean=round(float(itr.mean()), 2), iters_max=int(itr.max()),
                    status_counts={str(k_): int((str_ == k_).sum()) for k_ in np.unique(str_)}, f_mean=round(float(fr.mean()), 5),
                    chord_rad_5_50_95=[round(float(x_), 2) for x_ in np.percentile(chord_, [5, 50, 95])],
                    chord_rad_stored_5_50_95=[syn.STORED_CHORD_RAD[args.robot.split("_")[0]][i_] for i_ in (1, 4, 7)],
                    ms_one_call_host_api=round(1e3 * tr_, 3), trajectories_per_s_one_call=round(B / tr_, 1),
                    what="64 goal sets of 8 neighbouring grasps at the stored plans' joint-space distances (make_goal_sets_reference_shaped), "
                         "one solver call through the host-pointer API; thresholds 1 cm / 5 deg of examples/pybullet_gto_planning.py:262")
                rs_ = quality["reference_shaped"]
                # gated: limits, no plan in collision, cost not above the seed's; goal_ok_frac >= 0.95 is reported as met / not met
                # (the misses are rotation errors of 5-10 deg at positions within 5 mm: the optimum of the reference's objective)
                rs_["goal_ok_ge_0.95"] = bool(rs_["goal_ok_frac"] >= 0.95)
                rs_["gate"] = "pass" if (rs_["max_joint_limit_violation"] <= 1e-8 and rs_["plans_in_collision_frac"] == 0.0 and rs_["goal_pos_ok_frac"] >= 0.95) else "FAIL"
                gate_ok = gate_ok and rs_["gate"] == "pass"
                quality["_reference_shaped_args"] = argsr + (RTr, amr)  # (for the CPU port's twin; removed before the line is written)
        quality["gate"] = "pass" if gate_ok else "FAIL"
        if not gate_ok:
            rc = 3
        # what the gate looks at and what it does not (printed, not part of the exit code): the goal thresholds of
        # examples/pybullet_gto_planning.py:262 (1 cm / 5 deg) are missed by about a fifth of the SINGLE-goal instances --
        # a property of the reference's objective weights (velocity 0.01 / dt^2 against a goal term that is a sum of 200
        # squared point distances: the optimum trades a few millimetres of goal error for a shorter path), which the CPU
        # port shares (cpu_baseline.goal_ok_frac on the same batch) and L-BFGS-B confirmed in round 2
        # (profiles/r02_goal_miss_check.txt); with goal sets of eight (plan_goalset's call shape) 0.92 reach a goal
        quality["gate_reasons"] = {
            "checked": ["max_joint_limit_violation <= 1e-8", "objective_le_seed_frac == 1"] + ([] if args.light else [
                "stopping tolerance costs a converged instance <= 1e-4 of its objective", "plan_cost_le_seed_frac >= 0.95 (table top)",
                "central_diff leaves no more plans in collision than zero-gradient", "goal sets of 8 within the joint limits"]),
            "reported_not_gated": {"goal_ok_frac": quality["goal_ok_frac"],
                                   "goal_ok_frac_goal_sets_of_8": (quality.get("goal_sets_of_8") or {}).get("goal_ok_frac"),
                                   "cause": "objective weights of gto/gto_planner.py:84-135 (velocity term against point-matching goal term): the "
                                            "minimiser of the reference's own objective ends millimetres from a single goal; the CPU port ends at "
                                            "the same trajectories (cpu_baseline.goal_ok_frac, max_abs_dQ_vs_gpu)"}}

        # roofline of the dominant kernel.  SURVEY.md 8d prices the reference's work at 7 float32 (28 B) per surface point
        # and free waypoint; the broad phase proves most of those gathers to be exact zeros and skips them, so only the
        # points the kernel actually looked up (device counter) are priced: 28 algorithmic bytes each (the kernel reads one
        # 32-B voxel record per point instead of seven floats) over the HIP-event time of the kernel's launches.
        P = desc.n_points
        per_batch_ev = (iters + 1).reshape(M, B).sum(axis=1)  # each instance is evaluated iters+1 times per solve
        evals = float(sum(per_batch_ev[:m].sum() for m in plan))
        full_points = evals * (T - 2) * P
        avg_launch_us = 1e3 * kern_ms / max(kern_launches, 1)
        achieved = gathered * 28 / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = issue = None
        pmc_variants = {}
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):  # PMC figures are quoted only for the workload, mode and call size they were measured on
            for tr in json.load(open(tj)).get("entries", []):
                if (args.robot, args.grid, B * max(plan), slots, args.mode, bool(args.shelf)) == (tr.get("robot"), tr.get("grid"), tr.get("instances_per_call"),
                                                                                                    tr.get("slots"), tr.get("mode", "rounds"), bool(tr.get("shelf", False))):
                    traffic = tr.get("hbm_bytes_per_launch")
                    issue = tr.get("issue")
                    pmc_variants = tr.get("variants", {})
        # the same per kernel VARIANT: launches, HIP-event time, gathered points (binned by variant on the device) and PMC traffic
        # all belong to one population, so traffic / alg_bytes_per_launch is a valid ratio; the step kernels ride along
        loop_ms = sum(v_[0] for v_ in variants.values()) or 1.0
        by_variant = {}
        for name_, (ms_, nl_, wg_, pts_) in variants.items():
            if not nl_:
                continue
            row = {"launches": nl_, "avg_launch_us": round(1e3 * ms_ / nl_, 2), "workgroups_per_launch": round(wg_ / nl_, 1),
                   "share_of_solve_loop_kernel_time": round(ms_ / loop_ms, 4)}
            if name_.startswith("k_obstacle"):
                ach_ = pts_ * 28 / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0.0
                pm_ = pmc_variants.get(name_, {})
                row.update({"points_gathered_per_launch": round(pts_ / nl_), "alg_bytes_per_launch": round(pts_ * 28 / nl_),
                            "achieved": round(ach_, 1), "frac": round(ach_ / HBM_PEAK_GBS, 4),
                            "traffic": pm_.get("hbm_bytes_per_launch"),
                            "traffic_over_alg_bytes": round(pm_["hbm_bytes_per_launch"] / max(pts_ * 28 / nl_, 1.0), 2) if pm_.get("hbm_bytes_per_launch") else None})
            else:
                pm_ = pmc_variants.get(name_, {})
                row.update({"traffic": pm_.get("hbm_bytes_per_launch"), "fetch_bytes_per_launch": pm_.get("fetch_bytes_per_launch"),
                            "bound": "latency (serial chain of 8x8 block eliminations per instance: DESIGN.md section 5)"})
            pm_ = pmc_variants.get(name_, {})
            # issue side of the same PMC passes: vector instructions x 4 cycles over every SIMD's cycles of the launch, waves per
            # SIMD of the launch, share of the waves' lifetime spent waiting, LDS bank conflicts; and the stamped critical path
            # of one workgroup (GTO_DEBUG_TIMING, tools/step_stamps.py) over the launch
            for k_ in ("valu_issue_frac", "waves_per_simd", "waves_waiting_frac", "lds_bank_conflict_cycles_per_lds_inst", "l2_hit_rate",
                       "critical_path_cycles", "critical_path_over_launch"):
                if k_ in pm_:
                    row[k_] = pm_[k_]
            by_variant[name_] = row
        # the headline figures: the obstacle variant SURVEY.md 8d's byte model is about -- the one that gathers the bytes (the
        # launches that fill the GPU); which kernel takes the most time is dominant_by_time
        dominant = max((n_ for n_ in by_variant if n_.startswith("k_obstacle")), key=lambda n_: (variants[n_][3], variants[n_][0]), default=None)
        if dominant:  # the headline figures are the dominant VARIANT's own (one population of launches)
            kernel_name, dv = dominant, by_variant[dominant]
            achieved, traffic = dv["achieved"], dv["traffic"]
            avg_launch_us, kern_launches_hl, gathered_hl = dv["avg_launch_us"], dv["launches"], variants[dominant][3]
        else:
            kern_launches_hl, gathered_hl = kern_launches, gathered
        roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_unit": 28, "unit_of_work": "surface point looked up in a cost field (value + 6 neighbours, SURVEY.md 8d)",
                    "points_gathered_per_launch": round(gathered_hl / max(kern_launches_hl, 1)),
                    "alg_bytes_per_launch": round(gathered_hl * 28 / max(kern_launches_hl, 1)),
                    "alg_bytes_skipped_frac": round(1.0 - gathered / max(full_points, 1.0), 4),
                    "avg_launch_us": round(avg_launch_us, 2), "launches": kern_launches_hl,
                    "variants": by_variant,
                    # the kernel with the largest share of the solve loop's kernel time, whatever bounds it (the
                    # step kernel, latency / issue bound: no byte model applies, its issue-side fractions are in its row)
                    "dominant_by_time": (lambda n_: dict(by_variant[n_], kernel=n_))(max(by_variant, key=lambda n_: by_variant[n_]["share_of_solve_loop_kernel_time"])) if by_variant else None,
                    # the timed regime (all lanes in flight, where launches of different lanes stretch each other): the bytes
                    # gathered by the region's calls over the region's wall time
                    "timed_regime": {"alg_bytes_region": int(gathered * 28), "region_ms": round(1e3 * elapsed, 3),
                                     "achieved": round(gathered * 28 / elapsed / 1e9, 1), "frac": round(gathered * 28 / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                                     "one_lane_kernel_ms_per_step": round(loop_ms / max(prof_steps, 1), 4),
                                     "what": "points gathered by the region's solver calls (counted while one lane repeats them) x 28 B over the "
                                             "median timed region; one_lane_kernel_ms_per_step = kernel time of the solve loop per step with ONE "
                                             "lane on the GPU (it exceeds ms_per_step when the lanes overlap)"},
                    # what the counters mean here (profiles/r06_counter_calibration.txt, tools/probes/gather32_probe.hip): one memory-side
                    # read request fetches a 128-B line and FETCH_SIZE tallies it at 64 B, for 32-B random gathers as for streaming
                    # reads (x 2 holds); WRITE_SIZE is exact for whole 64-B lines.  A COLD 32-B record gather therefore moves
                    # 128 B per 28 algorithmic bytes: traffic_over_alg_bytes 4.57 means "every lookup a line of its own", below it
                    # the records are re-used out of L2 / Infinity Cache
                    "traffic_calibration": "FETCH_SIZE x 2 + WRITE_SIZE; cold 32-B gathers: 4.57 x alg (profiles/r06_counter_calibration.txt)",
                    "instances_per_call": B * max(plan), "slots": slots,
                    # the kernel is bound by instruction issue and latency, not by HBM: the same PMC passes give the share of the
                    # time the FP64 vector units are busy (a second roofline axis: 1.0 = every SIMD issuing vector work every cycle)
                    "valu_fp64_frac": None if not issue else issue.get("fp64_valu_busy_frac"),
                    "waves_waiting_frac": None if not issue else issue.get("waves_waiting_frac"),
                    "lds_bank_conflict_cycles_per_lds_inst": None if not issue else issue.get("lds_bank_conflict_cycles_per_lds_inst"),
                    "issue": issue,
                    "measured": "HIP events on the launch stream around every launch of the kernel while ONE lane repeats the timed "
                                "region's solver calls right after it, so that launches of other lanes do not stretch the durations; "
                                "points gathered: device counter of the same launches; traffic / issue: rocprofv3 PMC passes of the "
                                "same call size (profiles/traffic.json)"}

        cpu_baseline = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a 1-GPU-run item (rank 0, N = 1)
            from oracle import oracle
            oracle.build()
            o = oracle.Oracle(desc, cfg["link_ee"], cfg["link_gripper"], opts, n_gripper_points=100)
            o.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
            cores = o.usable_cores()  # affinity mask capped by the cgroup CPU quota
            # pilot on the batch itself, then repeat it so the timed sample is ~10-20 s of CPU work
            tc = time.perf_counter()
            cq, cr, cs_, cb, c0 = qc[:B], RT[:B].reshape(B, 1, 16), S[:B], base[:B], Q0[:B]  # the first batch
            Qo, _, fo, ito, sto = o.solve_batch(0, cq, cr, 1, cs_, cb, c0, n_threads=cores)
            t_pilot = time.perf_counter() - tc
            reps = int(min(max(np.ceil(args.cpu_seconds / max(t_pilot, 1e-3)), 1), 64))
            tile = lambda a: np.concatenate([a] * reps)
            tc = time.perf_counter()
            _, _, _, ito_all, _ = o.solve_batch(0, tile(cq), tile(cr), 1, tile(cs_), tile(cb), tile(c0), n_threads=cores)
            tcpu = time.perf_counter() - tc
            ns = B * reps
            # one thread, a few instances: the per-core rate without the machine's other cores competing for memory
            n1 = int(min(max(round(4.0 * (B / max(t_pilot, 1e-3)) / cores), 2), 8))
            t1 = time.perf_counter()
            o.solve_batch(0, cq[:n1], cr[:n1], 1, cs_[:n1], cb[:n1], c0[:n1], n_threads=1)
            t1 = time.perf_counter() - t1
            cpu_baseline = {"value": round(ns / tcpu, 3), "unit": "trajectories/s", "cores": cores, "kind": "port",
                            "sample": f"{reps} x the {B} instances of this workload ({ns} solves), {tcpu:.1f} s, "
                                      "OpenMP over instances, same algorithm in FP64 (oracle/gto_oracle.c); cores = CPUs this "
                                      f"process may use (affinity and cgroup quota) of {os.cpu_count()} hardware threads on the host",
                            "iters_per_s": round(float(ito_all.sum()) / tcpu, 1),
                            "single_core_value": round(ns / tcpu / cores, 4),
                            "single_thread": {"value": round(n1 / t1, 4), "sample": f"the first {n1} instances on one thread, {t1:.1f} s"},
                            "max_abs_dQ_vs_gpu": float(np.abs(Qo - Qsol[:B]).max()),
                            "iters_equal_gpu": bool(np.array_equal(ito, iters[:B])),
                            # the quality block of the CPU port on the same 64 instances, next to the GPU's on them: what the
                            # goal misses are a property of (objective and weights) shows in both
                            "quality": quality_block(desc, cfg, h, 0, Qo, RT[:B], Q0[:B], ito.astype(np.int64), sto, args.max_iter),
                            "quality_gpu_same_instances": quality_block(desc, cfg, h, 0, Qsol[:B], RT[:B], Q0[:B], iters[:B], status[:B], args.max_iter)}
            rsa = quality.get("_reference_shaped_args")
            if rsa is not None:  # the reference-shaped workload on the CPU port: same 64 goal sets, same seeds
                Qro, _, fro, itro, stro = o.solve_batch(*rsa[:7], n_threads=cores)
                _, _, _, amo = o.eval_objective(0, rsa[2], rsa[3], S[0].reshape(4, 4), [0.0, 0.0, 0.0], Qro)
                qbo = quality_block(desc, cfg, h, 0, Qro, rsa[7][np.arange(B), amo], rsa[6], itro.astype(np.int64), stro, args.max_iter)
                quality["reference_shaped"].update(goal_ok_frac_cpu_port=qbo["goal_ok_frac"], iters_mean_cpu_port=round(float(itro.mean()), 2),
                                                   cpu_port_same_goals_reached=bool(np.array_equal(amo, rsa[8])))
        quality.pop("_reference_shaped_args", None)

        # ---- oracle spot check (--oracle-check n): the first n instances of the first batch again on the CPU oracle
        oracle_check = None
        if args.oracle_check > 0:
            from oracle import oracle
            oracle.build()
            o_ = oracle.Oracle(desc, cfg["link_ee"], cfg["link_gripper"], opts, n_gripper_points=100)
            o_.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
            nc_ = min(args.oracle_check, B)
            tco = time.perf_counter()
            Qo_, _, fo_, ito_, sto_ = o_.solve_batch(0, qc[:nc_], RT[:nc_].reshape(nc_, 1, 16), 1, S[:nc_], base[:nc_], Q0[:nc_], n_threads=min(nc_, o_.usable_cores()))
            oracle_check = {"instances": nc_, "iters_equal": bool(np.array_equal(ito_, iters[:nc_])), "status_equal": bool(np.array_equal(sto_, status[:nc_])),
                            "max_abs_dQ_vs_oracle": float(np.abs(Qo_ - Qsol[:nc_]).max()),
                            "max_rel_dcost_vs_oracle": float(np.max(np.abs(fo_ - cost[:nc_]) / np.maximum(np.abs(fo_), 1e-300))),
                            "iters": ito_.tolist(), "seconds": round(time.perf_counter() - tco, 2)}
            if not (oracle_check["iters_equal"] and oracle_check["max_abs_dQ_vs_oracle"] < 1e-6):
                quality["gate"] = "FAIL"
                rc = 3

        # ---- BASELINE configs[2] and [4] on the same record (N = 1 only): each a short run of this script's own code path
        # (four lanes, calls of 8 steps), with its own roofline and an oracle spot check on four instances
        other_configs = None
        if world == 1 and not child and not args.no_other_configs and not args.no_cpu_baseline and args.robot == "panda_5k" and not args.shelf:
            other_configs = {}
            common = ["--gpus", "1", "--light", "--no-cpu-baseline", "--no-next-rows", "--no-other-configs", "--repeats", "3", "--warmup", "1",
                      "--oracle-check", "4", "--pipeline", str(args.pipeline)]
            for key_, argv_ in (("configs[2] fetch_shelf_256", ["--robot", "fetch", "--batch", "256", "--shelf", "--merge", "8", "--steps", "32"]),
                                ("configs[4] fetch_mobile_T80_256^3", ["--robot", "fetch_mobile", "--T", "80", "--grid", "256", "--shelf", "--batch", "64",
                                                                      "--merge", "8", "--steps", "32"])):
                t_oc = time.perf_counter()
                try:
                    r_ = main(common + argv_, emit=False)
                    other_configs[key_] = {"argv": " ".join(argv_), "trajectories_per_s": r_["value"], "ms_per_step": r_["ms_per_step"],
                                           "timed_regions": r_["timed_regions"], "workload": r_["config"]["workload"],
                                           "iters_mean": r_["iters_mean"], "iters_max": r_["iters_max"], "status_counts": r_["status_counts"],
                                           "roofline": {k_: r_["roofline"][k_] for k_ in ("kernel", "achieved", "frac", "traffic", "points_gathered_per_launch",
                                                                                         "alg_bytes_per_launch", "alg_bytes_skipped_frac", "avg_launch_us", "launches",
                                                                                         "timed_regime", "issue", "dominant_by_time")},
                                           "kernel_variants": r_["roofline"]["variants"],
                                           "oracle_check": r_["oracle_check"], "gate": r_["quality"]["gate"],
                                           "seconds": round(time.perf_counter() - t_oc, 2)}
                    if r_["quality"]["gate"] != "pass":
                        rc = 3
                except SystemExit as e_:  # the child's gate failed
                    other_configs[key_] = {"argv": " ".join(argv_), "error": f"exit {e_.code}"}
                    rc = 3

        # ---- the rows SURVEY.md 8(f) marks "next" and one plan_goalset call, with an oracle spot check each (N = 1 only)
        next_rows = None
        if not args.no_cpu_baseline and world == 1 and not args.no_next_rows:
            import importlib.util
            spec = importlib.util.spec_from_file_location("gto_next_rows", os.path.join(ROOT, "tools", "next_rows.py"))
            nr = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(nr)
            t_nr = time.perf_counter()
            next_rows = nr.measure(device=local_rank, scene=sc if (args.grid == 128 and not fetch and not args.shelf) else None)
            next_rows["seconds"] = round(time.perf_counter() - t_nr, 2)

        out = {
            "metric": "grasp trajectories/sec", "value": round(value, 2), "unit": "trajectories/s",
            # which of the two rates `value` is: the device-resident entry point (inputs and outputs in HBM when the timed region
            # starts and ends); SURVEY.md 8d's literal metric -- the host-pointer entry point with the per-instance H2D / D2H
            # inside the timing -- is host_api.trajectories_per_s on the same line, measured on the same lanes and regions
            "metric_definition": {"value": "gto_solve_batch_device: per-instance inputs and outputs resident in HBM (timed region = K steps, barrier + "
                                           "synchronize on both sides; median of `timed_regions.repeats` regions)",
                                  "survey_8d_literal": "host_api.trajectories_per_s: gto_solve_batch with host pointers, per-instance H2D / D2H inside the timed "
                                                       "region, one-time scene upload excluded"},
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            # the timed region (exactly `steps` steps, barrier + synchronize on both sides) is run `repeats` times; value and
            # ms_per_step are the median region
            "timed_regions": {"repeats": R, "what": "median", "ms_per_step_all": [round(1e3 * float(x) / args.steps, 3) for x in el_all],
                              "value_min": round(total_traj / float(el_all.max()), 1), "value_max": round(total_traj / float(el_all.min()), 1),
                              "spread_rel": round(float((el_all.max() - el_all.min()) / np.median(el_all)), 4)},
            # SURVEY.md 8d's literal metric, same regions through the host-pointer entry point (H2D / D2H of the per-instance
            # data inside the timing)
            "host_api": None if host_all is None else {
                "trajectories_per_s": round(world * B * args.steps / float(np.median(host_all)), 2),
                "ms_per_step": round(1e3 * float(np.median(host_all)) / args.steps, 3),
                "ms_per_step_all": [round(1e3 * float(x) / args.steps, 3) for x in host_all],
                "vs_device_resident": round(float(np.median(el_all) / np.median(host_all)), 4),
                "spread_rel": round(float((host_all.max() - host_all.min()) / np.median(host_all)), 4)},
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[4]: Fetch on a planar base ({desc.n_opt} optimised joints), 1 scene x {B} goal grasps per GPU, T={int(T)}, " if mobile else
                                    f"BASELINE configs[2]: Fetch arm, {'shelf' if args.shelf else 'table-top'} scene x {B} goal grasps per GPU, T={int(T)}, " if fetch else
                                    f"BASELINE configs[1]: Panda 7-DoF, 1 scene x {B} goal grasps per GPU, T={int(T)}, ") +
                                   f"{P} surface points, {args.grid}^3 f32 SDF cost field",
                       "batch_per_gpu": B, "T": int(T), "surface_points": int(P), "grid": args.grid,
                       "max_iter": args.max_iter, "steps_per_call": M, "lanes": D, "solver_mode": args.mode,
                       "parallelism": f"instances sharded over {world} GPU(s), no collective"
                                      + (f" (dry run: {world} ranks on ONE device, backend {args.backend})" if args.same_device else "")},
            "pipeline": {"lanes": D, "steps_per_call": M, "slots_per_lane": slots,
                         "calls_in_steps": sorted(set(plan)),
                         "what": "per GPU: `lanes` solver handles (HIP stream + host thread each, every lane its own grasp sets); every "
                                 "solver call of a lane gets `steps_per_call` consecutive steps' batches and keeps at most "
                                 "`slots_per_lane` of their instances in flight, handing a finished instance's slot to the next",
                         "serial_ms_per_step": None if args.merged_launches_only else round(1e3 * serial_elapsed / args.steps, 3),
                         "serial_trajectories_per_s": None if args.merged_launches_only else round(B * args.steps / serial_elapsed, 2),
                         "host_cpu_cores_busy": round(host_cpu / elapsed, 2),
                         # trajectories/s of the same K steps with 1 / 2 / all lanes (median of 3 regions; the last is `value`)
                         "lane_rates": lane_rates,
                         "lane_results_reproducible_alone": lanes_reproducible, "merged_equals_single_batch_solves": merged_equals_single},
            "sqp_iters_per_s": round(iters_per_s, 1),
            # SURVEY.md 8d's metric: gto_solve_batch with host pointers, H2D / D2H of the per-instance data inside the timing,
            # through the same lanes and call sizes as the timed region; and one batch per call
            "host_api_pipelined_trajectories_per_s": None if host_pipe_rate is None else round(world * host_pipe_rate, 2),
            "host_api_trajectories_per_s": None if host_rate is None else round(host_rate, 2),
            "iters_mean": round(float(iters.mean()), 2), "iters_max": int(iters.max()),
            "status_counts": {str(k): int((status == k).sum()) for k in np.unique(status)},
            "quality": quality,
            "reference_published": "0.098 trajectories/s (Panda tabletop, IPOPT on unknown CPU; BASELINE.md section 1)",
            "roofline": roofline, "cpu_baseline": cpu_baseline, "scene_sharded": scene_sharded, "next_rows": next_rows,
            "other_configs": other_configs, "oracle_check": oracle_check, "collective": collective,
            "lib_sha16": lib_sha16(),
        }
        if emit:
            emit_lines(out)
    for ln in reversed(lanes):  # the owner of the shared scene goes last
        ln.h.close()
    if multi and not child:
        dist.destroy_process_group()
    if rc and not child:
        raise SystemExit(rc)
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
