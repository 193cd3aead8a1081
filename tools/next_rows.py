#!/usr/bin/env python3
"""next_rows.py — the rows SURVEY.md 8(f) marks "next" (the callers and data formats either side of the solve), measured
on the GPU box with an oracle spot check each, for the `next_rows` object of bench.py's JSON line (N = 1, rank 0):

  f-1  batched IK pre-filter (gto/ik_solver.py:30-110, examples/pybullet_gto_planning.py:242-272 `ik_time`): IK/s with and
       without the collision term, 1024 goal poses per call
  f-2  cost field from a depth image (mesh_to_sdf/depth_point_cloud.py:9-141, examples/pybullet_gto_planning.py:181-190):
       ms for a 480x640 image at 48^3 and 128^3 voxels
  f-3  seed scoring (gto/gto_models.py:204-215, gto/gto_planner.py:197-213): plan scores/s
  f-4  base placement (gto/base_planner.py:35-168): goal sets/s, 1024 sets x 10 goals
  a1   ONE GTOPlanner.plan_goalset call end to end through the drop-in surface (examples/pybullet_gto_planning.py:290-294
       `planning_time`): ms

Every number is tied to a correct result: `check` compares 8 samples of the timed call's output with the CPU oracle
(oracle/gto_oracle.c: test infrastructure, here the checker, never the thing measured).
usage: python tools/next_rows.py   (prints the object as JSON)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _timed(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def _cfg(name):
    return json.load(open(os.path.join(ROOT, "grasptrajopt_amd", "data", f"{name}_cfg.json")))


def test_image(H=480, W=640):
    """A floor seen at an angle with a few boxes on it, one millimetre of sensor noise (tools/depth_field_rate.py)."""
    rng = np.random.default_rng(0)
    K = np.array([[600.0, 0, 320.0], [0, 600.0, 240.0], [0, 0, 1.0]])
    v, u = np.mgrid[0:H, 0:W]
    depth = (1.0 + 0.0012 * (v - H / 2) + 0.0003 * (u - W / 2)).astype(np.float32)
    for (r0, r1, c0, c1, dz) in ((150, 260, 200, 330, 0.2), (280, 400, 380, 520, 0.1), (100, 180, 420, 480, 0.3)):
        depth[r0:r1, c0:c1] -= dz
    depth += (0.001 * rng.standard_normal((H, W))).astype(np.float32)
    a = 0.5
    cam = np.eye(4)
    cam[:3, :3] = np.array([[0, -np.sin(a), np.cos(a)], [-1.0, 0, 0], [0, -np.cos(a), -np.sin(a)]])
    cam[:3, 3] = [-0.3, 0.0, 0.9]
    return depth, K, cam


def measure(device=0, scene=None, n_ik=1024, n_sets=1024, n_goals_base=10, n_plans=2048, check=8):
    import grasptrajopt_amd as g
    from grasptrajopt_amd import _capi, synthetic as syn
    from grasptrajopt_amd.robot_desc import load_builtin
    from oracle import oracle
    oracle.build()
    out = {"what": "rows SURVEY.md 8(f) marks next + one plan_goalset call, measured here with an oracle spot check each "
                   f"(max difference to the CPU port on {check} samples of the timed call's result)"}
    cfg = _cfg("panda")
    desc = load_builtin("panda_5k")
    opts = _capi.default_opts()
    h = _capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=device, n_gripper_points=100)
    orc = oracle.Oracle(desc, cfg["link_ee"], cfg["link_gripper"], opts, n_gripper_points=100)
    sc = scene if scene is not None else syn.make_scene(0, n=128, res=2.24 / 128)
    for s in (h, orc):
        s.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    sel = np.linspace(0, n_ik - 1, check).astype(int)

    # ---- f-1 IK pre-filter
    RT, _ = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], n_ik, seed=0)
    q0 = np.tile(np.array(cfg["default_pose"]), (n_ik, 1))
    base0 = np.zeros((n_ik, 3))
    fe = desc.frame_index(cfg["link_ee"])
    ik = {"reference": "gto/ik_solver.py:30-110; timer ik_time, examples/pybullet_gto_planning.py:244,270", "goals_per_call": n_ik}
    for key, sid in (("without_collision_term", None), ("with_collision_term", 0)):
        dt, (q, f, it, st) = _timed(lambda: h.solve_ik_batch(sid, q0, RT.reshape(n_ik, 16), base0), 3)
        qo, fo, ito, sto = orc.solve_ik_batch(sid, q0[sel], RT[sel].reshape(-1, 16), base0[sel])
        Tf = h.eval_fk(q)[:, fe]
        ep = np.linalg.norm(Tf[:, :3, 3] - RT[:, :3, 3], axis=1)
        ik[key] = {"ik_per_s": round(n_ik / dt, 1), "ms_per_call": round(1e3 * dt, 3), "iters_mean": round(float(it.mean()), 2),
                   "reached_1cm_frac": round(float((ep < 0.01).mean()), 3),
                   "check": {"max_abs_dq_vs_oracle": float(np.abs(q[sel] - qo).max()), "iters_equal": bool(np.array_equal(it[sel], ito)),
                             "status_equal": bool(np.array_equal(st[sel], sto))}}
    # the 1024-goal call is latency-bound (it ends with its slowest goal, 50 iterations); four times the goals per call show
    # what the kernel sustains once the GPU has more than one pass of workgroups to run
    n_big = 4 * n_ik
    RTb, _ = syn.make_goals(desc, h.eval_fk, cfg["link_ee"], n_big, seed=0)
    q0b, base0b = np.tile(np.array(cfg["default_pose"]), (n_big, 1)), np.zeros((n_big, 3))
    dtb, (qb, fb, itb, stb) = _timed(lambda: h.solve_ik_batch(0, q0b, RTb.reshape(n_big, 16), base0b), 3)
    selb = np.linspace(0, n_big - 1, check).astype(int)
    qob, fob, itob, stob = orc.solve_ik_batch(0, q0b[selb], RTb[selb].reshape(-1, 16), base0b[selb])
    ik["with_collision_term_4x_goals"] = {"goals_per_call": n_big, "ik_per_s": round(n_big / dtb, 1), "ms_per_call": round(1e3 * dtb, 3),
                                          "iters_mean": round(float(itb.mean()), 2),
                                          "check": {"max_abs_dq_vs_oracle": float(np.abs(qb[selb] - qob).max()),
                                                    "iters_equal": bool(np.array_equal(itb[selb], itob)), "status_equal": bool(np.array_equal(stb[selb], stob))}}
    out["f1_ik"] = ik

    # ---- f-3 seed scoring
    rng = np.random.default_rng(0)
    lo, hi = desc.lower[desc.opt_index], desc.upper[desc.opt_index]
    Qp = np.zeros((n_plans, desc.ndof, opts.T))
    Qp[:, desc.opt_index, :] = rng.uniform(lo[None, :, None], hi[None, :, None], (n_plans, len(lo), opts.T))
    dt, (c, d_) = _timed(lambda: h.plan_cost(0, Qp, [0.0, 0.0, 0.0]), 3)
    selp = np.linspace(0, n_plans - 1, check).astype(int)
    co, do = orc.plan_cost(0, Qp[selp], [0.0, 0.0, 0.0])
    out["f3_plan_scores"] = {"reference": "gto/gto_models.py:204-215 compute_plan_cost, gto/gto_planner.py:197-213", "plans_per_call": n_plans,
                             "plan_scores_per_s": round(n_plans / dt, 1), "ms_per_call": round(1e3 * dt, 3),
                             "point_lookups_per_s": round(n_plans * opts.T * desc.n_points / dt, 0),
                             "check": {"max_rel_dcost_vs_oracle": float(np.max(np.abs(c[selp] - co) / np.maximum(np.abs(co), 1e-300))),
                                       "max_abs_ddist_vs_oracle": float(np.abs(d_[selp] - do).max())}}
    h.close()

    # ---- f-2 cost field from a depth image
    depth, K, cam = test_image()
    dpc = g.DepthPointCloud(depth, K, cam)
    f2 = {"reference": "mesh_to_sdf/depth_point_cloud.py:9-141 (KD-tree in the reference); examples/pybullet_gto_planning.py:181-190",
          "image": "480x640, %d valid pixels" % int((depth > 0).sum())}
    for n in (48, 128):
        ax = np.linspace(-0.4, 1.84, n)
        qv = np.stack(np.meshgrid(ax, ax - 0.72, ax, indexing="ij"), -1).reshape(-1, 3)
        dt, c = _timed(lambda: dpc.get_sdf_cost(qv), 3)
        selq = np.concatenate([np.nonzero(c > 0)[0][:check // 2], np.linspace(0, qv.shape[0] - 1, check - check // 2).astype(int)])
        ref = oracle.depth_sdf_cost(depth, K, cam, None, dpc.threshold, qv[selq])
        ref_cost = ref[3]  # (points, sdf, inside, cost)
        f2[f"grid_{n}"] = {"voxels": int(qv.shape[0]), "ms": round(1e3 * dt, 3), "nonzero_cost_voxels": int((c > 0).sum()),
                           "check": {"cost_bit_identical_to_oracle": bool(np.array_equal(np.asarray(c)[selq], np.asarray(ref_cost, dtype=np.float32)))}}
    out["f2_depth_cost_field"] = f2

    # ---- f-4 base placement
    cfgf = _cfg("fetch")
    df = load_builtin("fetch")
    hf = _capi.SolverHandle(df, cfgf["link_ee"], cfgf["link_gripper"], opts, device=device, n_gripper_points=100)
    of = oracle.Oracle(df, cfgf["link_ee"], cfgf["link_gripper"], opts, n_gripper_points=100)
    qcf = np.array(cfgf["default_pose"], dtype=np.float64)
    goals, _ = syn.make_base_goal_sets(df, hf.eval_fk, cfgf["link_ee"], qcf, n_sets, n_goals_base, 0)
    QC = np.tile(qcf, (n_sets, 1))
    sels = np.linspace(0, n_sets - 1, check).astype(int)
    f4 = {"reference": "gto/base_planner.py:35-168, examples/pybullet_gto_planning_mobile.py:183-199", "sets_per_call": n_sets, "goals_per_set": n_goals_base}
    for key, w in (("effort_weight_0", 0.0), ("effort_weight_0.01", 0.01)):
        dt, (y, q, c, it, st) = _timed(lambda: hf.solve_base_batch(QC, goals, effort_weight=w), 2)
        yo, qo, co, ito, sto = of.solve_base_batch(QC[sels], goals[sels], effort_weight=w)
        f4[key] = {"sets_per_s": round(n_sets / dt, 1), "ms_per_call": round(1e3 * dt, 3), "iters_mean": round(float(it.mean()), 2),
                   "check": {"max_rel_dcost_vs_oracle": float(np.max(np.abs(c[sels] - co) / np.maximum(np.abs(co), 1e-12))),
                             "max_abs_dy_vs_oracle": float(np.abs(y[sels] - yo).max()), "iters_equal": bool(np.array_equal(it[sels], ito))}}
    hf.close()
    out["f4_base_placement"] = f4

    # ---- a1 one plan_goalset call end to end through the drop-in surface
    robot = g.GTORobotModel(desc=g.load_builtin("panda_5k"), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                            collision_link_names=cfg["collision_link_names"], device=device)
    robot.grid_resolution = 0.0175
    rngp = np.random.default_rng(3)
    robot.setup_points_field(rngp.uniform([-0.72, -0.72, -0.02], [0.72, 0.72, 1.4], size=(2000, 3)))
    wp = robot.workspace_points
    d_table = wp[:, 2] - 0.0
    qb = np.abs(wp - np.array([0.55, 0.1, 0.1])) - np.array([0.06, 0.06, 0.1])
    d_box = np.linalg.norm(np.maximum(qb, 0), axis=1) + np.minimum(qb.max(axis=1), 0)
    c_all = syn.sdf_cost_map(np.minimum(d_table, d_box), epsilon=0.06).astype(np.float32)
    c_obs = syn.sdf_cost_map(d_table, epsilon=0.06).astype(np.float32)
    planner = g.GTOPlanner(robot, cfg["link_ee"], cfg["link_gripper"], standoff_distance=-0.1, standoff_offset=-10)
    n_goals = 64
    RTg, qsol = syn.make_goals(robot.desc, robot._util_handle().eval_fk, cfg["link_ee"], n_goals, seed=11, zlim=(0.15, 0.6))
    qc = np.array(cfg["default_pose"])
    q_solutions = qsol.T.astype(np.float32)
    call = lambda: planner.plan_goalset(qc, RTg, c_all, c_obs, [0.0, 0.0, 0.0], q_solutions, use_standoff=True,
                                        axis_standoff=cfg["axis_standoff"], interpolate=True)
    ts = []
    for r in range(8):
        t0 = time.perf_counter()
        plan, dQ, cost = call()
        ts.append(time.perf_counter() - t0)
    # the oracle on the same call: seed selection (plain sum of c_obs over the interpolated plans, lexsort) and the solve
    shape, origin, res = robot.field_geometry()
    op = oracle.Oracle(robot.desc, cfg["link_ee"], cfg["link_gripper"], opts)  # the planner's own gripper cloud: every point of the link
    op.set_scene(0, c_all, c_obs, shape, origin, res)
    plans = []
    for i in range(n_goals):
        data = syn.interpolate_waypoints(np.stack([qc, q_solutions.T.astype(np.float64)[i]]), 50, robot.ndof)
        data[:, robot.parameter_joint_indexes] = qc[robot.parameter_joint_indexes]
        plans.append(data.T)
    plans = np.stack(plans)
    pc, pd = op.plan_cost(0, plans, [0.0, 0.0, 0.0])
    best = int(np.lexsort((pd, pc))[0])
    S = syn.standoff_pose(-0.1, cfg["axis_standoff"])
    Qo, _, fo, ito, _ = op.solve_batch(0, qc[None], RTg.reshape(1, n_goals, 16), n_goals, S, [0.0, 0.0, 0.0], plans[best][None])
    out["a1_plan_goalset_call"] = {"reference": "gto/gto_planner.py:185-245; timer planning_time, examples/pybullet_gto_planning.py:290-294 (4-30 s published)",
                                   "what": f"field {tuple(int(x) for x in shape)} in, goal set of {n_goals}, seed chosen among {n_goals} IK solutions, one trajectory out",
                                   "ms_median": round(1e3 * float(np.median(ts[2:])), 3), "ms_min": round(1e3 * float(np.min(ts[2:])), 3),
                                   "iterations": int(planner.solver.number_of_iterations()),
                                   "check": {"seed_index_equal": bool(planner.seed_index == best), "max_abs_dQ_vs_oracle": float(np.abs(plan - Qo[0]).max()),
                                             "iters_equal": bool(planner.solver.number_of_iterations() == int(ito[0])),
                                             "rel_dcost": float(abs(float(cost[0]) - float(fo[0])) / max(abs(float(fo[0])), 1e-300))}}
    robot.close()
    # ---- the whole per-object pipeline through the drop-in surface (examples/pybullet_gto_planning.py:176-190, 242-272, 291):
    # device-resident since round 4, next to the host path of rounds 2-3 on the same inputs (same plan, checked)
    import importlib.util
    spec = importlib.util.spec_from_file_location("gto_pipeline_latency", os.path.join(ROOT, "tools", "pipeline_latency.py"))
    pl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pl)
    dv, ho = pl.stage_table(0.05, 64, 6, host=False, device=device), pl.stage_table(0.05, 64, 3, host=True, device=device)
    out["per_object_pipeline"] = {"reference": "examples/pybullet_gto_planning.py:176-190 (clouds, grid, two cost fields), :242-272 (IK of the grasps), :291 (plan_goalset); "
                                               "the reference's timers for these stages add up to 6-33 s per object",
                                  "device_resident": dv, "host_path_rounds_2_3": ho["ms"],
                                  "check": {"same_plan_cost_as_host_path": bool(dv["plan_cost"] == ho["plan_cost"]), "same_grid": dv["field_shape"] == ho["field_shape"]}}
    return out


if __name__ == "__main__":
    print(json.dumps(measure(), indent=1))
