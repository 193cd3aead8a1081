"""CPU: the oracle's building blocks against golden vectors produced by the reference's own
numerics (tests/golden/make_golden.py).  These pin the oracle."""
import json

import numpy as np
import pytest

from conftest import golden
from grasptrajopt_amd.robot_desc import load_builtin


@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_robot_table_matches_reference(robot, robot_cfgs):
    g = golden(f"fk_{robot}.npz")
    d = load_builtin(robot)
    assert d.ndof == int(g["ndof"])
    assert d.actuated_joint_names == [str(s) for s in g["actuated"]]
    assert d.opt_index.tolist() == g["opt_index"].tolist()
    assert d.param_index.tolist() == g["param_index"].tolist()
    np.testing.assert_array_equal(d.lower, g["lower"])
    np.testing.assert_array_equal(d.upper, g["upper"])
    np.testing.assert_array_equal(d.lower[d.opt_index], g["lower_opt"])
    np.testing.assert_array_equal(d.upper[d.opt_index], g["upper_opt"])
    assert d.link_names == [str(s) for s in g["visual_names"]]


@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_fk_frames_and_visual_tf(robot, robot_cfgs, oracle_mod):
    g = golden(f"fk_{robot}.npz")
    cfg = robot_cfgs[robot]
    d = load_builtin(robot)
    orc = oracle_mod.Oracle(d, cfg["link_ee"], cfg["link_gripper"])
    frames = orc.eval_fk(g["q"])
    names = [str(s) for s in g["link_names"]]
    for j, fn in enumerate(d.frame_names):
        ref = g["frames"][:, names.index(fn)]
        np.testing.assert_allclose(frames[:, j], ref, rtol=0, atol=2e-14, err_msg=fn)
    vis = orc.eval_visual_tf(g["q"])
    np.testing.assert_allclose(vis, g["visual"], rtol=0, atol=2e-14)
    # gripper_tf = invt(T_ee) @ T_gripper (gto/gto_planner.py:38)
    Te = frames[0, d.frame_index(cfg["link_ee"])]
    Tg = frames[0, d.frame_index(cfg["link_gripper"])]
    np.testing.assert_allclose(np.linalg.inv(Te) @ Tg, g["gripper_tf"], atol=1e-14)


def test_known_answer_panda_default_pose(oracle_mod, robot_cfgs):
    # default pose -> panda_hand at (0.146093, 0, 0.705968)
    d = load_builtin("panda")
    cfg = robot_cfgs["panda"]
    orc = oracle_mod.Oracle(d, cfg["link_ee"], cfg["link_gripper"])
    T = orc.eval_fk(np.array(cfg["default_pose"]))[0, d.frame_index("panda_hand")]
    np.testing.assert_allclose(T[:3, 3], [0.146093424, 0.0, 0.705968301], atol=1e-8)
    np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-15)


def test_rotation_primitives(oracle_mod):
    g = golden("fk_panda.npz")  # rpy2r/angvec2r are exercised inside FK; spot-check orthonormality
    for rpy in ([0.1, -0.7, 2.0], [-1.57079632679, 0, 0], [0, 0, 0]):
        R = oracle_mod.rpy2r(rpy)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-15)
    R = oracle_mod.angvec2r(0.3, [0, 0, 2.0])
    np.testing.assert_allclose(R, [[np.cos(0.3), -np.sin(0.3), 0], [np.sin(0.3), np.cos(0.3), 0], [0, 0, 1]], atol=1e-16)
    assert np.array_equal(oracle_mod.rpy2r([0, 0, 0]), np.eye(3))
    del g


def test_sdf_value_jac_hess(oracle_mod):
    g = golden("sdf_callback.npz")
    val, jac, hes = oracle_mod.sdf_eval(g["data"], g["shape"], g["origin"], float(g["res"]), g["points"])
    np.testing.assert_array_equal(val, g["value"])          # nearest-voxel value: exact
    np.testing.assert_allclose(jac, g["jac"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(hes, g["hess"], rtol=1e-14, atol=1e-12)


def test_points_to_offsets_exact(oracle_mod):
    g = golden("grid.npz")
    off = oracle_mod.points_to_offsets(g["query"], g["origin"], 0.05, g["field_shape"])
    np.testing.assert_array_equal(off, g["offsets"])
    # the sdf_callback index (floor+clip, gto/sdf_callback.py:38-40) agrees with it
    s = golden("sdf_callback.npz")
    off2 = oracle_mod.points_to_offsets(s["points"], s["origin"], float(s["res"]), s["shape"])
    np.testing.assert_array_equal(s["data"][off2].astype(np.float64), s["value"])


def test_cost_map(oracle_mod):
    g = golden("depth_cost.npz")
    for tag in ("all", "obs"):
        cost = oracle_mod.sdf_cost_map(g[f"{tag}_sdf"], g[f"{tag}_inside"], 0.02, 1.0)
        np.testing.assert_array_equal(cost, g[f"{tag}_cost"])


def test_interpolate_waypoints(oracle_mod):
    g = golden("interp.npz")
    for qc, qg, s50, s7 in zip(g["qc"], g["qgoal"], g["seeds"], g["seeds7"]):
        np.testing.assert_allclose(oracle_mod.interpolate_waypoints(np.stack([qc, qg]), 50, 9), s50, rtol=0, atol=2e-15)
        np.testing.assert_allclose(oracle_mod.interpolate_waypoints(np.stack([qc, qg]), 7, 9), s7, rtol=0, atol=2e-15)


def test_stored_plan_invariants():
    g = golden("stored_plans.npz")
    stats = json.loads(str(g["stats_json"]))
    assert sum(v["n"] for v in stats.values()) == 853
    for v in stats.values():
        assert v["max_q1_q0"] <= 1.3e-8   # zero initial velocity + Euler dynamics
    d = load_builtin("panda")
    plans = g["panda_tabletop_sample"]
    o = d.opt_index
    viol = np.maximum(d.lower[o, None] - plans[:, o], plans[:, o] - d.upper[o, None]).max()
    assert viol <= 1.0e-8 + 1e-12
    assert np.ptp(plans[:, d.param_index, :], axis=2).max() == 0.0


def test_ik_oracle_recovers_reachable_poses(oracle_mod):
    """IK restatement (gto/ik_solver.py:30-110): goal poses generated by FK of in-limit configurations are
    reached from a nearby seed; the objective at the solution is the collision term alone; a goal-free
    field only shifts the value, never the minimiser, in GTO_GRAD_ZERO mode (what CasADi AD sees)."""
    from helpers import Problem
    prob = Problem("panda", B=6, scene_seed=2)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], oracle_mod.reference_opts())
    prob.finish(o.eval_fk)
    o.set_scene(*prob.scene_args())
    rng = np.random.default_rng(0)
    q_goal = prob.qgoal[:, 0]
    q0 = q_goal.copy()
    oi = prob.desc.opt_index
    q0[:, oi] += rng.uniform(-0.2, 0.2, size=(6, len(oi)))
    q0[:, oi] = np.clip(q0[:, oi], prob.desc.lower[oi], prob.desc.upper[oi])
    q, f, it, st = o.solve_ik_batch(None, q0, prob.goals[:, 0], max_iter=50, n_threads=1)
    fe = prob.desc.frame_index(prob.cfg["link_ee"])
    Tf = o.eval_fk(q)[:, fe]
    RT = prob.goals[:, 0].reshape(6, 4, 4)
    assert np.abs(Tf[:, :3, 3] - RT[:, :3, 3]).max() < 1e-5
    assert np.abs(Tf[:, :3, :3] - RT[:, :3, :3]).max() < 1e-4
    assert (f < 1e-8).all() and (st == 0).all() and (it <= 50).all()
    # parameter joints are returned as given, optimised joints stay inside their limits
    pi = prob.desc.param_index
    np.testing.assert_array_equal(q[:, pi], q0[:, pi])
    assert (q[:, oi] >= prob.desc.lower[oi]).all() and (q[:, oi] <= prob.desc.upper[oi]).all()
    # collision term with zero gradient: same minimiser, objective shifted by w * sum(c)
    oz = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], oracle_mod.reference_opts(grad_mode=1))
    oz.set_scene(*prob.scene_args())
    qz, fz, itz, _ = oz.solve_ik_batch(0, q0, prob.goals[:, 0], prob.base, max_iter=50, n_threads=1)
    np.testing.assert_allclose(qz, q, atol=1e-4)  # the constant shift moves the relative-decrease stop a little
    _, _, val, _ = oz.eval_points(0, qz, prob.base, use_obs=True)
    np.testing.assert_allclose(fz, 10.0 * val.sum(axis=1), rtol=1e-6, atol=1e-7)


def test_depth_cost_field_against_reference_golden(oracle_mod):
    """Row f-2, pinned: back-projection, KD-tree distance, is_outside and the cost map of the reference's
    DepthPointCloud (mesh_to_sdf/depth_point_cloud.py), generated by running the reference itself
    (tests/golden/make_golden.py); the restatement reproduces every value bit for bit."""
    g = golden("depth_cost.npz")
    for tag, tm in (("all", None), ("obs", g["mask"])):
        pts, sdf, inside, cost = oracle_mod.depth_sdf_cost(g["depth"], g["K"], g["cam"], tm, 1.5, g[f"{tag}_query"])
        np.testing.assert_array_equal(pts, g[f"{tag}_points"])
        np.testing.assert_array_equal(sdf, g[f"{tag}_sdf"])
        np.testing.assert_array_equal(inside, g[f"{tag}_inside"])
        np.testing.assert_array_equal(cost, g[f"{tag}_cost"])
    assert g["all_inside"].any() and (~g["all_inside"]).any()


def test_base_placement_oracle_recovers_reachable_goal_sets(oracle_mod):
    """Base-placement restatement (gto/base_planner.py:35-94): goal sets generated as reachable from a
    displaced base are matched to zero residual without the effort term (Gauss-Newton converges in a few
    iterations, which pins the Jacobian of every block); with the effort term the optimum trades a small
    residual for a smaller base motion; a single goal needs no base motion at all."""
    from helpers import cfg_of
    from grasptrajopt_amd import load_builtin, synthetic as syn
    cfg, desc = cfg_of("fetch"), load_builtin("fetch")
    o = oracle_mod.Oracle(desc, cfg["link_ee"], cfg["link_gripper"], oracle_mod.reference_opts(), n_gripper_points=100)
    qc = np.array(cfg["default_pose"], dtype=np.float64)
    B, n = 4, 6
    goals, ystar = syn.make_base_goal_sets(desc, o.eval_fk, cfg["link_ee"], qc, B, n, seed=1)
    QC = np.tile(qc, (B, 1))
    y, q, f, it, st = o.solve_base_batch(QC, goals, effort_weight=0.0, n_threads=1)
    assert (f < 1e-8).all() and (st == 0).all() and (it <= 60).all()  # relative-decrease stop: tol_rel_f = 1e-8
    fe, fg = desc.frame_index(cfg["link_ee"]), desc.frame_index(cfg["link_gripper"])
    fr = o.eval_fk(q.reshape(B * n, -1)).reshape(B, n, -1, 4, 4)
    for b in range(B):  # tf_base @ RT_i == FK_ee(q_i): the base pose and the arm postures are consistent
        np.testing.assert_allclose(syn.base_pose_matrix(y[b]) @ goals[b], fr[b, :, fe], atol=1e-4)
    oi, pi = desc.opt_index, desc.param_index
    assert (q[..., oi] >= desc.lower[oi] - 1e-12).all() and (q[..., oi] <= desc.upper[oi] + 1e-12).all()
    np.testing.assert_array_equal(q[..., pi], np.broadcast_to(qc[pi], q[..., pi].shape))
    assert (np.abs(y[:, 2]) <= np.pi).all()
    # effort term: smaller base motion, small positive residual; objective below the zero-effort point's value
    yw, qw, fw, _, _ = o.solve_base_batch(QC, goals, effort_weight=0.01, n_threads=1)
    assert (np.linalg.norm(yw, axis=1) < np.linalg.norm(y, axis=1)).all()
    assert (fw <= 0.01 * (y ** 2).sum(axis=1) + 1e-9).all()
    # ragged goal sets: a set's solution does not depend on the padding rows
    ng = np.array([n, 3, 1, 2], dtype=np.int32)
    yr, qr, fr_, itr, _ = o.solve_base_batch(QC, goals, n_goals=ng, effort_weight=0.0, n_threads=1)
    np.testing.assert_array_equal(yr[0], y[0])
    y1, q1, f1, _, _ = o.solve_base_batch(QC[1:2], goals[1:2, :3], effort_weight=0.0, n_threads=1)
    np.testing.assert_array_equal(yr[1], y1[0])
    np.testing.assert_array_equal(qr[1, :3], q1[0])
    np.testing.assert_array_equal(qr[1, 3:], np.broadcast_to(qc, (n - 3, len(qc))))


def test_occupancy_grid_against_reference_golden():
    """x-y occupancy grid of the mobile pipeline (gto/gto_models.py:218-273), pinned on a fixture produced
    by the reference's own setup_occupancy_grid / points_to_offsets_occupancy_numpy (sklearn KD-tree)."""
    import types
    from grasptrajopt_amd.gto_models import GTORobotModel
    d = golden("occupancy.npz")
    o = types.SimpleNamespace(field_margin=0.4, grid_resolution=0.05)
    GTORobotModel.setup_occupancy_grid(o, d["cloud"])
    np.testing.assert_array_equal(o.occupancy_grid_origin, d["origin"])
    assert tuple(o.occupancy_grid_shape) == tuple(d["shape"]) and o.occupancy_grid_size == int(d["size"])
    np.testing.assert_array_equal(o.occupancy_grid, d["grid"])
    assert 0 < o.occupancy_grid.sum() < o.occupancy_grid_size
    np.testing.assert_array_equal(GTORobotModel.points_to_offsets_occupancy_numpy(o, d["query"]), d["offsets"])
