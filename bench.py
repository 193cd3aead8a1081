#!/usr/bin/env python3
"""bench.py — throughput of the GTO inner solve on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: solving B independent (scene, goal-grasp)
trajectory problems of BASELINE.json configs[1] (Panda 7-DoF, T=50 waypoints, ~5k surface points,
128^3 float32 cost field, 64 candidate goal grasps of one scene) with the batched Gauss-Newton/LM
solver behind the C ABI (gto_solve_batch_device: inputs already resident in HBM).
N GPUs = N ranks, each solving its own scene x 64 grasps (weak scaling, no data-path collective).
The K timed steps run through grasptrajopt_amd.parallel.BatchPipeline with --pipeline D steps in flight
per GPU (one solver handle + stream + host thread each): a single batch of 64 is a latency-bound chain
of launches that leaves most CUs idle, and consecutive batches are independent.  The strictly serial
rate (D = 1) is reported next to it in "pipeline".

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (k_obstacle_gram), algorithmic field-gather bytes / HIP-event time
  cpu_baseline  the CPU oracle (same algorithm, FP64, OpenMP over instances) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def plan_calls(n, lanes, merge):
    """Split n steps (batches) into solver calls of at most `merge` batches, a multiple of `lanes` calls when n allows,
    sizes within one of each other: every lane gets the same amount of work and exactly n steps are run."""
    if n <= 0:
        return []
    L = max(-(-n // merge), min(n, lanes))
    if L % lanes and n >= lanes * (L // lanes + 1):
        L = lanes * (L // lanes + 1)
    return [n // L + (1 if i < n % L else 0) for i in range(L)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="goal grasps (instances) per GPU per step")
    ap.add_argument("--pipeline", type=int, default=4, help="lanes per GPU: solver handles, each with its stream and host thread")
    ap.add_argument("--merge", type=int, default=32, help="steps (batches) a lane hands to the solver in one call; the solver keeps "
                    "GTO_SLOTS (384) of their instances in flight and refills slots as instances finish")
    ap.add_argument("--max-iter", type=int, default=100, help="iteration cap (reference IPOPT cap: 100)")
    ap.add_argument("--robot", default="panda_5k")
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--merged-launches-only", action="store_true",
                    help="skip the one-batch-per-launch passes (serial latency, host API): every launch of the run then has the "
                         "timed region's size, which is what the per-launch PMC averages of tools/pmc_pass.sh need")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="target duration of the CPU-baseline sample")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (see profiles/)")
    args = ap.parse_args()

    # the HIP runtime multiplexes streams onto this many hardware queues (default 4); the pipeline lanes
    # must not share one, or their launches serialise
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the GTO solve path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as g
    if not os.path.exists(g.HIP_LIB):
        g.build()
    from grasptrajopt_amd import _capi, synthetic as syn
    from grasptrajopt_amd.parallel import BatchPipeline, shard_range
    from grasptrajopt_amd.robot_desc import load_builtin

    fetch = args.robot.startswith("fetch")  # BASELINE configs[2]: --robot fetch --batch 256 (shelf-height table)
    cfg = json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", f"{args.robot.split('_')[0]}_cfg.json")))
    desc = load_builtin(args.robot)
    opts = _capi.default_opts()
    opts.max_iter = args.max_iter
    T, ndof, B = opts.T, desc.ndof, args.batch
    D, M = max(1, args.pipeline), max(1, args.merge)
    slots = int(os.environ.get("GTO_SLOTS", "384"))  # instances a solver call keeps in flight (gto_api.hip)

    # this rank's shard of the global problem list: scene = global rank id, 64 grasps each
    lo, hi = shard_range(world * B, rank, world)
    assert hi - lo == B
    scene_seed = lo // B
    res = 2.24 / args.grid  # covers the 2.24 m reach box (0.0175 m at 128^3)
    sc = syn.make_scene(scene_seed, n=args.grid, res=res, origin=(-0.3, -1.12, 0.0) if fetch else (-0.4, -1.12, -0.4),
                        table_z=0.45 if fetch else -0.03)

    # D pipeline lanes: each one solver handle bound to ONE stream of its own, with its own copy of M batches
    # (M consecutive steps, distinct goal sets) in HBM and its own outputs; a launch solves m <= M of them at
    # once (grasptrajopt_amd.parallel.BatchPipeline runs the lanes concurrently)
    class Lane:
        def __init__(self, first=None):
            self.stream = torch.cuda.Stream(dev)
            self.h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=local_rank, n_gripper_points=100)
            self.h.set_stream(self.stream.cuda_stream)
            if first is None:
                self.h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
            else:
                self.h.share_scene(0, first.h)  # one copy of the scene in HBM for all lanes

        def upload(self, qc, RT, S, base, Q0):
            n = qc.shape[0]  # M * B instances, batch m in rows [m B, (m+1) B)
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
            self.inp = [torch.zeros(n, dtype=torch.int32, device=dev), t(qc, torch.float64), t(RT.reshape(n, 1, 16), torch.float64),
                        torch.ones(n, dtype=torch.int32, device=dev), t(S, torch.float64), t(base, torch.float64), t(Q0, torch.float64)]
            self.d_Q = torch.empty((n, ndof, T), dtype=torch.float64, device=dev)
            self.d_dQ = torch.empty((n, ndof, T - 1), dtype=torch.float64, device=dev)
            self.d_cost = torch.empty(n, dtype=torch.float64, device=dev)
            self.d_it = torch.empty(n, dtype=torch.int32, device=dev)
            self.d_st = torch.empty(n, dtype=torch.int32, device=dev)
            self.bufs = self.inp + [self.d_Q, self.d_dQ, self.d_cost, self.d_it, self.d_st]

        def step(self, m=1, first=0):
            """One launch over batches first .. first+m-1 of this lane."""
            ptrs = [x.data_ptr() + first * B * x.stride(0) * x.element_size() for x in self.bufs]
            self.h.solve_batch_device(m * B, 1, *ptrs, self.stream.cuda_stream)

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
    sets = [syn.make_goals(desc, h.eval_fk, cfg["link_ee"], B, seed=scene_seed + 1009 * m, collision_cost=goal_collision_cost, zlim=zlim)
            for m in range(M)]  # M different grasp sets in the same scene: the batches of M consecutive steps
    RT, qg = np.concatenate([x[0] for x in sets]), np.concatenate([x[1] for x in sets])
    NB = M * B
    qc = np.tile(np.array(cfg["default_pose"]), (NB, 1))
    Q0 = np.stack([syn.make_seed(qc[b], qg[b], T, desc.param_index) for b in range(NB)])
    S = np.tile(syn.standoff_pose(-0.1, cfg["axis_standoff"]).reshape(1, 16), (NB, 1))
    base = np.zeros((NB, 3))
    for ln in lanes:
        ln.upload(qc, RT, S, base, Q0)
    torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    launch_plan = lambda n: plan_calls(n, D, M)

    def run_steps(pipe, n):
        futs = [pipe.submit("step", m) for m in launch_plan(n)]
        for f in futs:
            f.result()

    pipe = BatchPipeline(lanes)
    run_steps(pipe, args.warmup * D * M)  # every lane sees >= W warmup launches
    barrier()
    # ---- timed region: exactly K steps (batches), up to D x M of them in flight
    t0, c0 = time.perf_counter(), time.process_time()
    run_steps(pipe, args.steps)
    barrier()
    elapsed, host_cpu = time.perf_counter() - t0, time.process_time() - c0
    pipe.close()
    merged_Q, merged_it = lanes[0].d_Q.clone(), lanes[0].d_it.clone()
    same = all(bool(torch.equal(l.d_Q, merged_Q)) and bool(torch.equal(l.d_it, merged_it)) for l in lanes[1:])

    # ---- the same K steps strictly one after the other, one batch per launch, on one lane: per-batch
    # latency (and a check that a batch solved alone equals the batch solved inside a merged launch)
    ln0 = lanes[0]
    barrier()
    ts = time.perf_counter()
    merged_equals_single = None
    if not args.merged_launches_only:
        for k in range(args.steps):
            ln0.step(1, k % M)
        barrier()
        covered = min(args.steps, M) * B
        merged_equals_single = bool(torch.equal(ln0.d_Q[:covered], merged_Q[:covered])) and bool(torch.equal(ln0.d_it[:covered], merged_it[:covered]))
    serial_elapsed = time.perf_counter() - ts
    # ---- the launches of the timed region once more, alone on the GPU, with HIP events around every launch
    # of the dominant kernel (on its launch stream) for the roofline: a launch's duration is then the
    # kernel's own and matches the rocprofv3 summary
    h.set_profiling(True)
    kern_ms, kern_launches, prof_steps = 0.0, 0, 0
    plan = launch_plan(args.steps)
    for m in plan:
        ln0.step(m)
        ms, nl = h.last_kernel_time()
        kern_ms += ms
        kern_launches += nl
        prof_steps += m
    barrier()
    h.set_profiling(False)
    ln0.step(M)  # full outputs again for the quality gate
    barrier()
    d_it, d_st, d_cost, d_Q = ln0.d_it, ln0.d_st, ln0.d_cost, ln0.d_Q

    # the same solve through the host-pointer entry point (H2D/D2H of per-instance data included)
    host_rate = None
    if rank == 0 and not args.merged_launches_only:
        hs = max(2, min(5, args.steps))
        hargs = (0, qc[:B], RT[:B].reshape(B, 1, 16), 1, S[:B], base[:B], Q0[:B])
        h.solve_batch(*hargs)
        th = time.perf_counter()
        for _ in range(hs):
            h.solve_batch(*hargs)
        host_rate = B * hs / (time.perf_counter() - th)

    iters = d_it.cpu().numpy().astype(np.int64)
    status = d_st.cpu().numpy()
    cost = d_cost.cpu().numpy()
    Qsol = d_Q.cpu().numpy()
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    # iterations done inside the timed region: a launch of m steps solves this lane's batches 0 .. m-1
    per_batch_it = iters.reshape(M, B).sum(axis=1)
    it_timed = float(sum(per_batch_it[:m].sum() for m in plan))
    it_sum = torch.tensor([it_timed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(it_sum, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_traj = world * B * args.steps
    value = total_traj / elapsed
    iters_per_s = float(it_sum.item()) / elapsed

    if rank == 0:
        # quality gate on this rank's batch
        oi = desc.opt_index
        viol = float(np.maximum(desc.lower[oi][None, :, None] - Qsol[:, oi], Qsol[:, oi] - desc.upper[oi][None, :, None]).max())
        fe = desc.frame_index(cfg["link_ee"])
        Tf = h.eval_fk(Qsol[:, :, -1])[:, fe]
        err_pos = np.linalg.norm(Tf[:, :3, 3] - RT[:, :3, 3], axis=1)
        cosang = (np.einsum("bij,bij->b", Tf[:, :3, :3], RT[:, :3, :3]) - 1.0) / 2.0
        err_rot = np.degrees(np.arccos(np.clip(cosang, -1, 1)))
        seed_cost, _ = h.plan_cost(0, np.clip(Q0, None, None), [0, 0, 0])
        sol_cost, _ = h.plan_cost(0, Qsol, [0, 0, 0])

        # roofline of the dominant kernel: algorithmic bytes (7 float32 gathers per
        # surface point and free waypoint) / HIP-event time of its launches in the timed region
        P = desc.n_points
        bytes_per_inst_launch = (T - 2) * P * 28
        per_batch_ev = (iters + 1).reshape(M, B).sum(axis=1)  # each instance is evaluated iters+1 times per solve
        evals = float(sum(per_batch_ev[:m].sum() for m in plan))
        alg_bytes = evals * bytes_per_inst_launch
        avg_launch_us = 1e3 * kern_ms / max(kern_launches, 1)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = args.traffic
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if traffic is None and os.path.exists(tj):  # quoted only for the workload and launch size it was measured on
            tr = json.load(open(tj))
            if (args.robot, args.grid, B * max(plan), slots) == (tr.get("robot"), tr.get("grid"), tr.get("instances_per_call"), tr.get("slots")):
                traffic = tr.get("k_obstacle_gram_hbm_bytes_per_launch")
        roofline = {"bound": "hbm", "kernel": "k_obstacle_gram", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": round(alg_bytes / max(kern_launches, 1)),
                    "avg_launch_us": round(avg_launch_us, 2), "launches": kern_launches,
                    "instances_per_call": B * max(plan), "slots": slots,
                    "measured": "HIP events on the launch stream around every launch of the kernel while ONE lane repeats the timed "
                                "region's solver calls right after it, so that launches of other lanes do not stretch the durations"}

        cpu_baseline = None
        if not args.no_cpu_baseline:
            from oracle import oracle
            oracle.build()
            o = oracle.Oracle(desc, cfg["link_ee"], cfg["link_gripper"], opts, n_gripper_points=100)
            o.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
            cores = o.usable_cores()  # affinity mask capped by the cgroup CPU quota
            # pilot on the batch itself, then repeat it so the timed sample is ~10-20 s of CPU work
            tc = time.perf_counter()
            cq, cr, cs_, cb, c0 = qc[:B], RT[:B].reshape(B, 1, 16), S[:B], base[:B], Q0[:B]  # the first batch
            Qo, _, fo, ito, _ = o.solve_batch(0, cq, cr, 1, cs_, cb, c0, n_threads=cores)
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
                            "iters_equal_gpu": bool(np.array_equal(ito, iters[:B]))}

        out = {
            "metric": "grasp trajectories/sec", "value": round(value, 2), "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[2]-like: Fetch arm, 1 scene x {B} goal grasps per GPU, T={int(T)}, " if fetch else
                                    f"BASELINE configs[1]: Panda 7-DoF, 1 scene x {B} goal grasps per GPU, T={int(T)}, ") +
                                   f"{P} surface points, {args.grid}^3 f32 SDF cost field",
                       "batch_per_gpu": B, "T": int(T), "surface_points": int(P), "grid": args.grid,
                       "max_iter": args.max_iter, "steps_per_call": M, "lanes": D,
                       "parallelism": f"instances sharded over {world} GPU(s), no collective"},
            "pipeline": {"lanes": D, "steps_per_call": M, "slots_per_lane": slots,
                         "calls_in_steps": sorted(set(plan)),
                         "what": "per GPU: `lanes` solver handles (HIP stream + host thread each); every solver call of a lane gets "
                                 "`steps_per_call` consecutive steps' batches (different grasp sets of the scene) and keeps at most "
                                 "`slots_per_lane` of their instances in flight, handing a finished instance's slot to the next",
                         "serial_ms_per_step": None if args.merged_launches_only else round(1e3 * serial_elapsed / args.steps, 3),
                         "serial_trajectories_per_s": None if args.merged_launches_only else round(B * args.steps / serial_elapsed, 2),
                         "host_cpu_cores_busy": round(host_cpu / elapsed, 2),
                         "lanes_bit_identical": same, "merged_equals_single_batch_solves": merged_equals_single},
            "sqp_iters_per_s": round(iters_per_s, 1),
            "host_api_trajectories_per_s": None if host_rate is None else round(host_rate, 2),
            "iters_mean": round(float(iters.mean()), 2), "iters_max": int(iters.max()),
            "status_counts": {str(k): int((status == k).sum()) for k in np.unique(status)},
            "quality": {"max_joint_limit_violation": viol, "goal_err_pos_max_m": round(float(err_pos.max()), 5),
                        "goal_err_rot_max_deg": round(float(err_rot.max()), 3),
                        "goal_ok_frac": round(float(((err_pos < 0.01) & (err_rot < 5)).mean()), 3),
                        "plan_cost_le_seed_frac": round(float((sol_cost <= seed_cost + 1e-12).mean()), 3),
                        "f_mean": round(float(cost.mean()), 5)},
            "reference_published": "0.098 trajectories/s (Panda tabletop, IPOPT on unknown CPU; BASELINE.md section 1)",
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    for ln in reversed(lanes):  # the owner of the shared scene goes last
        ln.h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
