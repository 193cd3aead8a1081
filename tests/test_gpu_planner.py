"""GPU: the drop-in GTOPlanner / GTORobotModel surface, called the way
examples/pybullet_gto_planning.py:124-141,179-190,291-293 calls the reference, checked against the
CPU oracle driven with the same inputs."""
import numpy as np
import pytest

from conftest import golden
import grasptrajopt_amd as g
from grasptrajopt_amd import synthetic as syn
from helpers import cfg_of

pytestmark = pytest.mark.gpu


def _setup(robot_name, oracle_mod, n_goals, rng):
    cfg = cfg_of(robot_name)
    robot = g.GTORobotModel(desc=g.load_builtin(robot_name), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                            collision_link_names=cfg["collision_link_names"], device=0)
    # scene point cloud -> 5 cm grid with 0.4 m margin, exactly as the driver does (:178-179)
    cloud = rng.uniform([0.25, -0.45, -0.02], [0.8, 0.45, 0.35], size=(400, 3))
    robot.setup_points_field(cloud)
    wp = robot.workspace_points
    # cost = band around a table slab and a box, through the reference's cost map
    d_table = wp[:, 2] - 0.0
    q = np.abs(wp - np.array([0.55, 0.1, 0.1])) - np.array([0.06, 0.06, 0.1])
    d_box = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)
    c_all = syn.sdf_cost_map(np.minimum(d_table, d_box), epsilon=0.06)
    c_obs = syn.sdf_cost_map(d_table, epsilon=0.06)
    planner = g.GTOPlanner(robot, cfg["link_ee"], cfg["link_gripper"], standoff_distance=-0.1, standoff_offset=-10)
    planner.max_iter = 30
    opts = oracle_mod.reference_opts(max_iter=30)
    orc = oracle_mod.Oracle(robot.desc, cfg["link_ee"], cfg["link_gripper"], opts)
    shape, origin, res = robot.field_geometry()
    orc.set_scene(0, c_all, c_obs, shape, origin, res)
    RT, qsol = syn.make_goals(robot.desc, orc.eval_fk, cfg["link_ee"], n_goals, seed=11, zlim=(0.15, 0.6))
    return cfg, robot, planner, orc, c_all, c_obs, RT, qsol


def _oracle_seed(robot, orc, qc, qsol, c_obs_scene, base, T=50):
    plans = []
    for i in range(qsol.shape[0]):
        data = syn.interpolate_waypoints(np.stack([qc, qsol[i]]), T, robot.ndof)
        data[:, robot.parameter_joint_indexes] = qc[robot.parameter_joint_indexes]
        plans.append(data.T)
    plans = np.stack(plans)
    cost, dist = orc.plan_cost(0, plans, base)
    return plans, int(np.lexsort((dist, cost))[0])


@pytest.mark.parametrize("interpolate", [True, False])
def test_plan_goalset_matches_oracle(oracle_mod, interpolate):
    rng = np.random.default_rng(3)
    cfg, robot, planner, orc, c_all, c_obs, RT, qsol = _setup("panda", oracle_mod, 4, rng)
    qc = np.array(cfg["default_pose"])
    base = [0.0, 0.0, 0.0]
    q_solutions = qsol.T.astype(np.float32)  # (ndof, n) float32 like the driver
    plan, dQ, cost = planner.plan_goalset(qc, RT, c_all, c_obs, base, q_solutions, use_standoff=True,
                                          axis_standoff=cfg["axis_standoff"], interpolate=interpolate)
    assert plan.shape == (9, 50) and dQ.shape == (9, 49) and cost.shape == (1,)
    plans, best = _oracle_seed(robot, orc, qc, q_solutions.T.astype(np.float64), c_obs, base)
    assert planner.seed_index == best
    if interpolate:
        Q0 = plans[best]
    else:
        Q0 = np.tile(qc[:, None], (1, 50))
        Q0[:, 40:] = plans[best][:, 49:50]
    S = syn.standoff_pose(-0.1, cfg["axis_standoff"])
    Qo, dQo, fo, ito, sto = orc.solve_batch(0, qc[None], RT.reshape(1, 4, 16), 4, S, base, Q0[None])
    assert planner.solver.number_of_iterations() == int(ito[0])
    np.testing.assert_allclose(plan, Qo[0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(cost, fo, rtol=1e-7)
    # invariants of the reference's stored plans
    assert np.abs(plan[:, 1] - plan[:, 0]).max() == 0 and np.array_equal(plan[7:], np.tile(qc[7:, None], (1, 50)))
    robot.close()


def test_plan_single_goal_ignores_sdf_cost_all_like_the_reference(oracle_mod):
    rng = np.random.default_rng(4)
    cfg, robot, planner, orc, c_all, c_obs, RT, qsol = _setup("panda", oracle_mod, 1, rng)
    qc = np.array(cfg["default_pose"])
    plan, dQ, cost = planner.plan(qc, RT[0], c_obs, [0, 0, 0], qsol[0], use_standoff=True, axis_standoff="z")
    # reference quirk (gto/gto_planner.py:165-173): sdf_cost_all is never set -> zeros before the standoff
    shape, origin, res = robot.field_geometry()
    orc.set_scene(1, np.zeros_like(c_obs), c_obs, shape, origin, res)
    Q0 = syn.make_seed(qc, qsol[0], 50, robot.desc.param_index)
    Qo, _, fo, ito, _ = orc.solve_batch(1, qc[None], RT[:1].reshape(1, 1, 16), 1, syn.standoff_pose(-0.1, "z"), [0, 0, 0], Q0[None])
    np.testing.assert_allclose(plan, Qo[0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(cost, fo, rtol=1e-7)
    assert planner.solver.stats()["iter_count"] == int(ito[0])
    robot.close()


def test_fetch_planner_and_surface_points(oracle_mod):
    rng = np.random.default_rng(5)
    cfg = cfg_of("fetch")
    robot = g.GTORobotModel(desc=g.load_builtin("fetch"), time_derivs=[0, 1], param_joints=cfg["param_joints"], device=0)
    gk = golden("fk_fetch.npz")
    pts, nrm = robot.compute_fk_surface_points(gk["q"][0])
    d = robot.desc
    for l in range(d.n_links):
        V = gk["visual"][0, l]
        sel = d.point_link == l
        np.testing.assert_allclose(pts[sel], d.points[sel] @ V[:3, :3].T + V[:3, 3], atol=1e-13)
        np.testing.assert_allclose(nrm[sel], d.normals[sel] @ V[:3, :3].T, atol=1e-12)
    tfb = np.eye(4)
    tfb[:3, 3] = [1.0, 2.0, 0.5]
    pts2, _ = robot.compute_fk_surface_points(gk["q"][0], tf_base=tfb)
    np.testing.assert_allclose(pts2, pts + tfb[:3, 3], atol=1e-13)
    # compute_plan_cost through the GPU == oracle
    robot.setup_points_field(rng.uniform([0.3, -0.4, 0.4], [0.9, 0.4, 1.0], size=(300, 3)))
    field = (0.05 * rng.random(robot.field_size)).astype(np.float32)
    plan = np.tile(np.array(cfg["default_pose"])[:, None], (1, 50))
    plan[6] += np.linspace(0, 0.5, 50)
    cost, dist = robot.compute_plan_cost(plan, field, [0.0, 0.1, 0.0])
    orc = oracle_mod.Oracle(d, cfg["link_ee"], cfg["link_gripper"])
    shape, origin, res = robot.field_geometry()
    orc.set_scene(0, field, None, shape, origin, res)
    co, do = orc.plan_cost(0, plan[None], [0.0, 0.1, 0.0])
    np.testing.assert_allclose([cost, dist], [co[0], do[0]], rtol=1e-12)
    robot.close()


def test_ik_solver_surface_matches_oracle(oracle_mod):
    """IKSolver called the way examples/pybullet_gto_planning.py:142,245,260 calls the reference:
    solve_ik(q0 (ndof,1), RT, sdf_cost_obstacle, base_position) -> (q, err_pos, err_rot, cost); then the
    batched form over all candidate grasps, which then feed plan_goalset as q_solutions."""
    rng = np.random.default_rng(5)
    cfg, robot, planner, orc, c_all, c_obs, RT, qsol = _setup("panda", oracle_mod, 6, rng)
    ik = g.IKSolver(robot, cfg["link_ee"], cfg["link_gripper"], collision_avoidance=True)
    ik.setup_optimization()
    qc = np.array(cfg["default_pose"])
    base = np.zeros(3)
    q, err_pos, err_rot, cost = ik.solve_ik(qc.reshape(-1, 1), RT[0], c_obs, base)
    qo, fo, ito, sto = orc.solve_ik_batch(0, qc[None], RT[0].reshape(1, 16), base, max_iter=50)
    assert q.shape == (robot.ndof,)
    np.testing.assert_allclose(q, qo[0], atol=1e-6)
    fe = robot.desc.frame_index(cfg["link_ee"])
    Tf = orc.eval_fk(qo)[0, fe]
    assert err_pos == pytest.approx(np.linalg.norm(RT[0][:3, 3] - Tf[:3, 3]), abs=1e-7)
    assert 0.0 <= err_rot <= 180.0
    _, _, val, _ = orc.eval_points(0, qo, base, use_obs=True)
    assert cost == pytest.approx(val.sum(), rel=1e-9, abs=1e-12)
    np.testing.assert_allclose(ik.solve_fk(q), orc.eval_fk(q[None])[0, fe], atol=1e-12)
    # all candidates at once, then the planner consumes the solutions that meet the reference's thresholds
    qb, ep, er, cb, it, st = ik.solve_ik_batch(qc, RT, c_obs, base)
    qob, _, itob, stob = orc.solve_ik_batch(0, np.tile(qc, (6, 1)), RT.reshape(6, 16), base, max_iter=50)
    np.testing.assert_array_equal(it, itob)
    np.testing.assert_allclose(qb, qob, atol=1e-6)
    ok = (ep < 0.01) & (er < 5)  # examples/pybullet_gto_planning.py:262
    assert ok.any()
    plan, dQ, c = planner.plan_goalset(qc, RT[ok], c_all, c_obs, base, qb[ok].T.astype(np.float32))
    assert plan.shape == (robot.ndof, 50) and np.isfinite(c).all()
    # no collision term: sdf not needed
    ik2 = g.IKSolver(robot, cfg["link_ee"], cfg["link_gripper"], collision_avoidance=False)
    q2, ep2, er2, c2 = ik2.solve_ik(qc, RT[1])
    assert c2 == 0.0 and q2.shape == (robot.ndof,)
    robot.close()


def test_base_planner_surface_matches_oracle(oracle_mod):
    """BasePlanner called the way examples/pybullet_gto_planning_mobile.py:127,183-199 calls the reference:
    setup_occupancy_grid(points); setup_optimization(n, weight); plan_goalset(q0 (ndof,1), RTs) ->
    (plan (ndof,n), y, err_pos, err_rot, cost); the base pose then moves the robot and the goals."""
    cfg = cfg_of("fetch")
    robot = g.GTORobotModel(desc=g.load_builtin("fetch"), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                            collision_link_names=cfg["collision_link_names"], device=0)
    orc = oracle_mod.Oracle(robot.desc, cfg["link_ee"], cfg["link_gripper"], oracle_mod.reference_opts())
    qc = np.array(cfg["default_pose"], dtype=np.float64)
    n = 6
    goals, ystar = syn.make_base_goal_sets(robot.desc, orc.eval_fk, cfg["link_ee"], qc, 2, n, seed=4)
    # observed scene: a table far in front of the robot -> the robot's footprint is free before and after the move
    rng = np.random.default_rng(0)
    cloud = np.c_[rng.uniform(1.6, 2.2, 4000), rng.uniform(-0.6, 0.6, 4000), rng.uniform(0.02, 0.75, 4000)]
    robot.setup_occupancy_grid(cloud)
    bp = g.BasePlanner(robot, cfg["link_ee"], cfg["link_gripper"])
    bp.setup_optimization(n, 0.0)
    plan, y, err_pos, err_rot, cost = bp.plan_goalset(qc.reshape(-1, 1), goals[0])
    assert plan.shape == (robot.ndof, n) and y.shape == (3,) and err_pos.shape == (n,) and err_rot.shape == (n,)
    yo, qo, fo, _, _ = orc.solve_base_batch(qc[None], goals[:1], effort_weight=0.0)
    np.testing.assert_allclose(y, yo[0], atol=1e-6)
    np.testing.assert_allclose(plan, qo[0].T, atol=1e-6)
    assert (err_pos < 1e-3).all() and (err_rot < 0.1).all()  # reachable by construction
    assert cost == 0.0
    # an obstacle exactly where the robot stands after the move is counted by the occupancy statistic
    RT_inv = np.linalg.inv(syn.base_pose_matrix(y))
    foot, _ = robot.compute_fk_surface_points(qc, tf_base=RT_inv)
    robot.setup_occupancy_grid(np.r_[cloud, foot[foot[:, 2] > 0.05][::7]])
    assert bp.base_collision_cost(qc, y) > 0
    # both goal sets in one call, and with the effort term the base moves less
    Q2, y2, ep2, er2, it2, st2 = bp.plan_goalset_batch(qc, goals)
    np.testing.assert_allclose(y2[0], y, atol=1e-9)
    bp.setup_optimization(n, 0.01)
    _, yw, _, _, _ = bp.plan_goalset(qc, goals[0])
    assert np.linalg.norm(yw) < np.linalg.norm(y)
    robot.close()


def test_evaluator_collision_statistic_composes(oracle_mod):
    """examples/pybullet_evaluate_plans.py:219-233 through the drop-in classes: FK surface points of every
    waypoint + DepthPointCloud.get_sdf, batched (utils.plan_in_collision) and as the reference's loop."""
    from grasptrajopt_amd.utils import plan_in_collision
    rng = np.random.default_rng(9)
    cfg, robot, planner, orc, c_all, c_obs, RT, qsol = _setup("panda", oracle_mod, 2, rng)
    # a synthetic depth camera looking at the workspace from above: a table plane at z = 0 with a box on it
    H, W = 60, 80
    K = np.array([[70.0, 0, 40.0], [0, 70.0, 30.0], [0, 0, 1.0]])
    cam = np.eye(4)
    cam[:3, :3] = np.array([[0, -1.0, 0], [-1.0, 0, 0], [0, 0, -1.0]])  # optical axis straight down
    cam[:3, 3] = [0.5, 0.0, 1.2]
    depth = np.full((H, W), 1.2, dtype=np.float32)
    depth[20:40, 30:50] = 0.9  # a box 30 cm high
    dpc = g.DepthPointCloud(depth, K, cam)
    qc = np.array(cfg["default_pose"])
    plan = np.tile(qc[:, None], (1, 50))
    # swing joint 2 so that the arm dips into the table in the second half
    plan[1] = np.linspace(qc[1], 1.6, 50)
    hit, first, count = plan_in_collision(robot, dpc, plan, [0, 0, 0])
    ref_count = []
    for i in range(50):
        pts, _ = robot.compute_fk_surface_points(plan[:, i])
        ref_count.append(int(np.sum(dpc.get_sdf(pts) < 0)))
    np.testing.assert_array_equal(count, ref_count)
    assert hit and first == int(np.nonzero(np.array(ref_count) > 5)[0][0]) and count[0] <= 5
    robot.close()


def test_grasp_collision_filter_matches_reference_loop(oracle_mod):
    """Row f-3: the driver's grasp filter (examples/pybullet_gto_planning.py:203-219) batched
    (utils.grasp_collision_ratio: one get_sdf call for all grasps) against the reference's own loop of
    compute_fk_surface_points(q, tf_base=RT @ standoff) + get_sdf per grasp, through the drop-in classes."""
    from grasptrajopt_amd.utils import grasp_collision_ratio, filter_grasps
    cfg = cfg_of("panda")
    # any GTORobotModel serves as the gripper model; here the arm's own point set at a fixed configuration
    gripper = g.GTORobotModel(desc=g.load_builtin("panda"), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                              collision_link_names=cfg["collision_link_names"], device=0)
    H, W = 60, 80
    K = np.array([[70.0, 0, 40.0], [0, 70.0, 30.0], [0, 0, 1.0]])
    cam = np.eye(4)
    cam[:3, :3] = np.array([[0, -1.0, 0], [-1.0, 0, 0], [0, 0, -1.0]])
    cam[:3, 3] = [0.5, 0.0, 1.5]
    depth = np.full((H, W), 1.5, dtype=np.float32)
    depth[15:45, 25:55] = 1.0  # a 0.5 m block on the floor
    dpc = g.DepthPointCloud(depth, K, cam)
    rng = np.random.default_rng(1)
    n = 24
    RT = np.tile(np.eye(4), (n, 1, 1))
    RT[:, :3, 3] = np.c_[rng.uniform(0.0, 1.0, n), rng.uniform(-0.6, 0.6, n), rng.uniform(-0.3, 0.9, n)]
    ang = rng.uniform(-np.pi, np.pi, n)
    RT[:, 0, 0], RT[:, 0, 1], RT[:, 1, 0], RT[:, 1, 1] = np.cos(ang), -np.sin(ang), np.sin(ang), np.cos(ang)
    off = syn.standoff_pose(-0.1, "z")
    q = np.array(cfg["default_pose"])
    ratio = grasp_collision_ratio(gripper, dpc, RT, q, off)
    ref = np.empty(n)
    for i in range(n):
        pts, _ = gripper.compute_fk_surface_points(q, tf_base=RT[i] @ off)
        sdf = dpc.get_sdf(pts)
        ref[i] = np.sum(sdf < 0) / len(sdf)
    np.testing.assert_allclose(ratio, ref, atol=2.0 / gripper.desc.n_points)  # points exactly on the zero level may flip
    assert (ratio > 0.01).any() and (ratio <= 0.01).any()
    np.testing.assert_array_equal(filter_grasps(gripper, dpc, RT, q, off), (ratio > 0.01).astype(np.int32))
    gripper.close()


@pytest.mark.parametrize("robot_name", ["panda", "fetch"])
@pytest.mark.parametrize("interpolate", [True, False])
def test_plan_goalset_seed_vs_reference_fixture(robot_name, interpolate):
    """Seed construction, scoring and selection of plan_goalset against what the reference's own plan_goalset does
    (gto/gto_planner.py:185-236 executed with a recording solver, tests/golden/make_objective_golden.py): same winning
    IK solution, same scores (the reference sums float32), same seed handed to the solver, same parameters."""
    gz = golden("objective.npz")
    cfg = cfg_of(robot_name)
    robot = g.GTORobotModel(desc=g.load_builtin(robot_name), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                            collision_link_names=cfg["collision_link_names"], device=0)
    # grid geometry as setup_points_field would leave it (gto/gto_models.py:155-171)
    robot.origin = gz[f"{robot_name}_grid_origin"].reshape(1, 3)
    robot.grid_resolution = float(gz[f"{robot_name}_grid_res"])
    robot.field_shape = tuple(int(v) for v in gz[f"{robot_name}_grid_shape"])
    robot.field_size = int(np.prod(robot.field_shape))
    tag = f"{robot_name}_seed_interp{int(interpolate)}"
    planner = g.GTOPlanner(robot, cfg["link_ee"], cfg["link_gripper"])
    planner.max_iter = 0  # the solver hands back its (clipped) seed
    qc = gz[f"{robot_name}_qc"]
    plan, dQ, cost = planner.plan_goalset(qc, gz[tag + "_RTs"], gz[f"{robot_name}_field_sparse_all"],
                                          gz[f"{robot_name}_field_sparse_obs"], gz[tag + "_base"], gz[tag + "_q_solutions"],
                                          use_standoff=True, axis_standoff=cfg["axis_standoff"], interpolate=interpolate)
    assert planner.seed_index == int(gz[tag + "_index"])
    np.testing.assert_allclose(planner.seed_cost_all, gz[tag + "_cost_all"], rtol=2e-6)
    np.testing.assert_allclose(planner.seed_dist_all, gz[tag + "_dist_all"], rtol=1e-13)
    Q0 = gz[tag + "_Q0"]
    oi, pi = robot.optimized_joint_indexes, robot.parameter_joint_indexes
    np.testing.assert_allclose(planner.solver.x0[f"{robot.get_name()}/q/x"], Q0[oi], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(planner.solver._p_dict[f"{robot.get_name()}/q/p"], Q0[pi])
    np.testing.assert_array_equal(planner.solver._p_dict["tf_goal"], gz[tag + "_tf_goal"])
    # the reference passes exactly these; whatever else the problem declares stays at its zero default
    passed = {str(k) for k in gz[tag + "_param_keys"]}
    assert passed <= set(planner.solver._p_dict)
    assert all(not np.any(v) for k, v in planner.solver._p_dict.items() if k not in passed)
    # the returned plan is that seed with the first two waypoints pinned to qc (Q_1 = Q_0 = qc: initial state + zero
    # initial velocity, gto/gto_planner.py:59-72) and the rest clipped into the joint limits
    lo, hi = robot.desc.lower[oi][:, None], robot.desc.upper[oi][:, None]
    exp = Q0.copy()
    exp[oi] = np.clip(exp[oi], lo, hi)
    exp[:, 0] = exp[:, 1] = qc
    np.testing.assert_allclose(plan, exp, rtol=0, atol=1e-13)
    robot.close()


def _depth_scene_inputs():
    """A camera looking down at a table with two boxes, 480 x 640 (tools/pipeline_latency.py's image)."""
    H, W = 480, 640
    K = np.array([[600.0, 0, 320.0], [0, 600.0, 240.0], [0, 0, 1.0]])
    v, u = np.mgrid[0:H, 0:W]
    depth = (1.0 + 0.0012 * (v - H / 2) + 0.0003 * (u - W / 2)).astype(np.float32)
    for (r0, r1, c0, c1, dz) in ((150, 260, 200, 330, 0.2), (280, 400, 380, 520, 0.1)):
        depth[r0:r1, c0:c1] -= dz
    depth[5:9, 7:30] = 0.0  # a few invalid pixels
    target = np.zeros((H, W), np.uint8)
    target[150:260, 200:330] = 1
    a = 0.9
    cam = np.eye(4)
    cam[:3, :3] = np.array([[0, -np.sin(a), np.cos(a)], [-1.0, 0, 0], [0, -np.cos(a), -np.sin(a)]])
    cam[:3, 3] = [-0.1, 0.0, 0.9]
    return depth, K, cam, target


@pytest.mark.parametrize("res,pattern", [(0.05, "one_image"), (0.031, "one_image"), (0.05, "driver"), (0.04, "driver")])
def test_device_resident_depth_scene_equals_the_host_path(oracle_mod, res, pattern):
    """examples/pybullet_gto_planning.py:176-190, 242-272, 291 called the reference's way: two DepthPointCloud objects,
    setup_points_field(points), get_sdf_cost(workspace_points) twice, IK of the grasps, plan_goalset.  Left lazy, all
    of it stays on the GPU (ONE gto_scene_from_depth call: depth_scene.py); forced through numpy (the host path of rounds
    2-3: points, grid and fields as arrays) it must give the same grid, bit-identical fields and the same IK and plan.
    pattern "driver": the obstacle cloud is built as the driver builds it (:187-189), from a COPY of the image with the
    target's pixels pushed to the threshold; "one_image": both clouds from the same array."""
    cfg = cfg_of("panda")
    depth, K, cam, target = _depth_scene_inputs()
    thr = 2.0
    depth_obstacle = depth
    if pattern == "driver":
        depth_obstacle = depth.copy()
        depth_obstacle[target.astype(bool)] = thr
    n_goals = 12
    qc = np.array(cfg["default_pose"])
    out = {}
    for path in ("host", "device"):
        robot = g.GTORobotModel(desc=g.load_builtin("panda"), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                                collision_link_names=cfg["collision_link_names"], device=0)
        robot.grid_resolution = res
        builds = []
        uh = robot._util_handle()
        real_build = uh.scene_from_depth
        uh.scene_from_depth = lambda *a, **k: (builds.append(k.get("depth_obstacle") is not None), real_build(*a, **k))[1]
        dpc_all = g.DepthPointCloud(depth, K, cam, target_mask=None, threshold=thr)
        pts = dpc_all.points
        if path == "host":
            pts = np.asarray(pts)  # the array the reference's DepthPointCloud hands out
        robot.setup_points_field(pts)
        wp = robot.workspace_points
        if path == "host":
            assert isinstance(wp, np.ndarray)
        c_all = dpc_all.get_sdf_cost(wp)
        dpc_obs = g.DepthPointCloud(depth_obstacle, K, cam, target_mask=target, threshold=thr)
        c_obs = dpc_obs.get_sdf_cost(wp)
        if path == "device":
            assert not isinstance(c_all, np.ndarray) and not isinstance(c_obs, np.ndarray)  # nothing came to the host
        ik = g.IKSolver(robot, cfg["link_ee"], cfg["link_gripper"], collision_avoidance=True)
        RT, _ = syn.make_goals(robot.desc, robot._util_handle().eval_fk, cfg["link_ee"], n_goals, seed=11, zlim=(0.15, 0.6))
        q_ik, ep, er, cost_ik, it, st = ik.solve_ik_batch(qc, RT, c_obs, [0.0, 0.0, 0.0])
        # the other field handed to an entry point that reads "the obstacle field": it is that field that must be read
        q_ik_all, _, _, cost_ik_all, it_all, _ = ik.solve_ik_batch(qc, RT, c_all, [0.0, 0.0, 0.0])
        planner = g.GTOPlanner(robot, cfg["link_ee"], cfg["link_gripper"], standoff_distance=-0.1, standoff_offset=-10)
        planner.max_iter = 25
        plan, dQ, cost = planner.plan_goalset(qc, RT, c_all, c_obs, [0.0, 0.0, 0.0], q_ik.T.astype(np.float32),
                                              use_standoff=True, axis_standoff=cfg["axis_standoff"], interpolate=True)
        pc, pd_ = robot.compute_plan_cost(plan, c_obs, [0.0, 0.0, 0.0])
        pc_all_first, _ = robot.compute_plan_cost(plan, c_all, [0.0, 0.0, 0.0])
        if path == "device":
            assert len(builds) == 1, f"{len(builds)} builds of the resident scene for one object"
            assert builds[0] == (pattern == "driver")  # the obstacle image went up with the call iff it is another image
        out[path] = dict(shape=robot.field_geometry()[0], origin=robot.field_geometry()[1], bounds=np.asarray(robot.workspace_bounds),
                         wp=np.asarray(robot.workspace_points), c_all=np.asarray(c_all), c_obs=np.asarray(c_obs), q_ik=q_ik, it=it,
                         q_ik_all=q_ik_all, it_all=it_all, cost_ik_all=cost_ik_all, pc_all=pc_all_first,
                         plan=plan, cost=cost, seed=planner.seed_index, pc=pc, size=robot.field_size)
        robot.close()
    h_, d_ = out["host"], out["device"]
    assert h_["shape"] == d_["shape"] and h_["size"] == d_["size"]
    np.testing.assert_array_equal(h_["origin"], d_["origin"])
    np.testing.assert_array_equal(h_["bounds"], d_["bounds"])
    np.testing.assert_array_equal(h_["wp"], d_["wp"])          # numpy.arange's values, reproduced on the host side of the C call
    np.testing.assert_array_equal(h_["c_all"], d_["c_all"])    # bit-identical float32 fields
    np.testing.assert_array_equal(h_["c_obs"], d_["c_obs"])
    assert (h_["c_all"] != h_["c_obs"]).any()
    # the field of all pixels sees the target object: voxels inside it cost something there and nothing in the obstacle field
    assert ((d_["c_all"] > 0) & (d_["c_obs"] == 0)).sum() > 0
    np.testing.assert_array_equal(h_["q_ik"], d_["q_ik"])
    np.testing.assert_array_equal(h_["it"], d_["it"])
    np.testing.assert_array_equal(h_["q_ik_all"], d_["q_ik_all"])
    np.testing.assert_array_equal(h_["it_all"], d_["it_all"])
    np.testing.assert_array_equal(h_["cost_ik_all"], d_["cost_ik_all"])
    assert h_["seed"] == d_["seed"]
    np.testing.assert_array_equal(h_["plan"], d_["plan"])
    np.testing.assert_array_equal(h_["cost"], d_["cost"])
    assert h_["pc"] == d_["pc"]
    assert h_["pc_all"] == d_["pc_all"]


def test_resident_depth_scene_is_rebuilt_not_reused_across_images(oracle_mod):
    """Two objects, one robot model (the driver's loop, examples/pybullet_gto_planning.py:160-300): the second object's
    fields come from another image.  A field of the FIRST object used after the second object's scene was built (its
    handles hold pointers into the first build's buffers) must still be its own field: the resident scene is keyed by
    value on both clouds, borrowers share again when its generation has moved."""
    cfg = cfg_of("panda")
    depth, K, cam, target = _depth_scene_inputs()
    depth2 = depth.copy()
    depth2[300:380, 100:180] -= 0.15  # another box on the table
    qc = np.array(cfg["default_pose"])
    robot = g.GTORobotModel(desc=g.load_builtin("panda"), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                            collision_link_names=cfg["collision_link_names"], device=0)
    fields = []
    for d in (depth, depth2):
        dobs = d.copy()
        dobs[target.astype(bool)] = 2.0
        dpc_all = g.DepthPointCloud(d, K, cam, target_mask=None, threshold=2.0)
        robot.setup_points_field(dpc_all.points)
        wp = robot.workspace_points
        fields.append((dpc_all.get_sdf_cost(wp), g.DepthPointCloud(dobs, K, cam, target_mask=target, threshold=2.0).get_sdf_cost(wp), dpc_all))
    ik = g.IKSolver(robot, cfg["link_ee"], cfg["link_gripper"], collision_avoidance=True)
    RT, _ = syn.make_goals(robot.desc, robot._util_handle().eval_fk, cfg["link_ee"], 6, seed=5, zlim=(0.15, 0.6))
    (a1, o1, p1), (a2, o2, p2) = fields
    q2, _, _, c2, it2, _ = ik.solve_ik_batch(qc, RT, o2, [0.0, 0.0, 0.0])   # builds object 2's scene
    planner = g.GTOPlanner(robot, cfg["link_ee"], cfg["link_gripper"], standoff_distance=-0.1, standoff_offset=-10)
    planner.max_iter = 10
    plan2, _, f2 = planner.plan_goalset(qc, RT, a2, o2, [0.0, 0.0, 0.0], q2.T.astype(np.float32), axis_standoff=cfg["axis_standoff"])
    solver2 = planner.solver
    # object 1's obstacle field now: the resident scene is rebuilt from object 1's images (on object 1's grid)
    q1, _, _, c1, it1, _ = ik.solve_ik_batch(qc, RT, o1, [0.0, 0.0, 0.0])
    # ... and the planner's solver of object 2, which borrowed the scene of the build that is gone, shares again
    sol = solver2.solve()
    np.testing.assert_array_equal(sol[f"{robot.get_name()}/q"].toarray(), plan2)
    # the same calls with the fields as host arrays
    robot_h = g.GTORobotModel(desc=g.load_builtin("panda"), time_derivs=[0, 1], param_joints=cfg["param_joints"],
                              collision_link_names=cfg["collision_link_names"], device=0)
    ik_h = g.IKSolver(robot_h, cfg["link_ee"], cfg["link_gripper"], collision_avoidance=True)
    robot_h.setup_points_field(np.asarray(p2.points))
    q2h, _, _, c2h, it2h, _ = ik_h.solve_ik_batch(qc, RT, np.asarray(o2), [0.0, 0.0, 0.0])
    robot_h.setup_points_field(np.asarray(p1.points))
    q1h, _, _, c1h, it1h, _ = ik_h.solve_ik_batch(qc, RT, np.asarray(o1), [0.0, 0.0, 0.0])
    np.testing.assert_array_equal(q2, q2h)
    np.testing.assert_array_equal(q1, q1h)
    np.testing.assert_array_equal(it1, it1h)
    assert (np.asarray(o1) != np.asarray(o2)).any() or np.asarray(o1).shape != np.asarray(o2).shape
    robot.close()
    robot_h.close()
