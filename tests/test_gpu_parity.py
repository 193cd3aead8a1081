"""GPU: parity of the HIP path (through the C ABI) with the CPU oracle and the golden fixtures.

Tolerances (stated here once):
  * integer work (voxel offsets, arg-min goal, iteration counts, status): bit-exact;
  * nearest-voxel cost values: exact (same float32 promoted to float64);
  * FK / points / gradients: 1e-12 absolute (FP64 on both sides; libm vs ocml sin/cos and FMA
    contraction differ by a few ulp);
  * Gauss-Newton blocks and objective terms: 1e-10 relative (different summation order);
  * solved trajectories after the same number of iterations: 1e-6 rad (north_star / SURVEY.md 8d).
"""
import numpy as np
import pytest

from conftest import golden
from grasptrajopt_amd import synthetic as syn
from helpers import Problem, cfg_of, point_cloud_robot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from grasptrajopt_amd import _capi
    return _capi


def make_pair(capi, oracle_mod, prob, mode=0, **opt_kw):
    opts = oracle_mod.reference_opts(**opt_kw)
    h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    h.set_mode(mode)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    prob.finish(h.eval_fk)
    h.set_scene(*prob.scene_args())
    o.set_scene(*prob.scene_args())
    return h, o


# ------------------------------------------------------------------------------------------ FK
@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_fk_matches_golden_and_oracle(capi, oracle_mod, robot):
    g = golden(f"fk_{robot}.npz")
    cfg = cfg_of(robot)
    from grasptrajopt_amd.robot_desc import load_builtin
    d = load_builtin(robot)
    h = capi.SolverHandle(d, cfg["link_ee"], cfg["link_gripper"], device=0)
    frames = h.eval_fk(g["q"])
    names = [str(s) for s in g["link_names"]]
    for j, fn in enumerate(d.frame_names):
        np.testing.assert_allclose(frames[:, j], g["frames"][:, names.index(fn)], rtol=0, atol=1e-13, err_msg=fn)
    o = oracle_mod.Oracle(d, cfg["link_ee"], cfg["link_gripper"])
    np.testing.assert_allclose(frames, o.eval_fk(g["q"]), rtol=0, atol=1e-13)
    # world surface points = visual_tf @ p (gto/gto_models.py:104-121) against the golden visual_tf
    xyz, _, _, _ = h.eval_points(0, g["q"][:4], [0.1, -0.2, 0.3], want_field=False)
    for i in range(4):
        for l in range(d.n_links):
            V = g["visual"][i, l]
            p = d.points[d.point_link == l]
            ref = p @ V[:3, :3].T + V[:3, 3] + np.array([0.1, -0.2, 0.3])
            np.testing.assert_allclose(xyz[i][d.point_link == l], ref, rtol=0, atol=1e-13)
    h.close()


# ------------------------------------------------------------------------------------------ field lookups
def test_field_lookup_against_reference_golden(capi, oracle_mod):
    """Voxel offsets (bit-exact), values (exact) and sdf_callback Jacobians on the reference's own
    golden vectors, incl. points exactly on voxel faces and outside the grid."""
    s = golden("sdf_callback.npz")
    d = point_cloud_robot(s["points"])
    h = capi.SolverHandle(d, "root", "root", device=0)
    h.set_scene(0, s["data"], None, s["shape"], s["origin"], float(s["res"]))
    xyz, off, val, grad = h.eval_points(0, np.zeros((1, 1)), [0, 0, 0])
    np.testing.assert_array_equal(xyz[0], s["points"])
    np.testing.assert_array_equal(val[0], s["value"])
    np.testing.assert_allclose(grad[0], s["jac"], rtol=2e-15, atol=0)
    # HesFun.eval (gto/sdf_callback.py:165-183) on the same points, incl. the clipped samples at the grid's border
    hess = h.eval_points_hessian(0, np.zeros((1, 1)), [0, 0, 0])
    np.testing.assert_allclose(hess[0], s["hess"], rtol=2e-15, atol=0)
    _, _, ho = oracle_mod.sdf_eval(s["data"], s["shape"], s["origin"], float(s["res"]), s["points"])
    np.testing.assert_array_equal(hess[0], ho.reshape(-1, 3, 3))
    ref_off = oracle_mod.points_to_offsets(s["points"], s["origin"], float(s["res"]), s["shape"])
    np.testing.assert_array_equal(off[0], ref_off)
    h.close()

    g = golden("grid.npz")
    d = point_cloud_robot(g["query"])
    h = capi.SolverHandle(d, "root", "root", device=0)
    F = int(g["field_size"])
    h.set_scene(5, np.arange(F, dtype=np.float32) % 1000, None, g["field_shape"], g["origin"].ravel(), 0.05)
    _, off, _, _ = h.eval_points(5, np.zeros((1, 1)), [0, 0, 0])
    np.testing.assert_array_equal(off[0], g["offsets"])  # gto/gto_models.py:190-201 executed by the reference
    h.close()


def test_points_offsets_values_vs_oracle(capi, oracle_mod):
    prob = Problem("panda", B=3, scene_seed=2, n=40, res=0.03, base=(0.02, -0.01, 0.03))  # small grid: many clipped points
    h, o = make_pair(capi, oracle_mod, prob)
    q = np.concatenate([prob.Q0[0].T[::7], prob.Q0[1].T[::9]])
    for use_obs in (False, True):
        xg, og, vg, gg = h.eval_points(0, q, prob.base[0], use_obs=use_obs)
        xo, oo, vo, go = o.eval_points(0, q, prob.base[0], use_obs=use_obs)
        np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-13)
        np.testing.assert_array_equal(og, oo)
        np.testing.assert_array_equal(vg, vo)
        np.testing.assert_allclose(gg, go, rtol=1e-15, atol=0)
    h.close()


# ------------------------------------------------------------------------------------------ objective pieces
@pytest.mark.parametrize("robot,n_goals,standoff", [("panda", 1, True), ("panda", 5, True), ("panda", 3, False),
                                                     ("fetch", 2, True)])
@pytest.mark.parametrize("mode", [0])
def test_objective_terms_vs_oracle(capi, oracle_mod, robot, n_goals, standoff, mode):
    prob = Problem(robot, B=5, scene_seed=3, n_goals=n_goals, use_standoff=standoff, base=(0.01, 0.0, -0.02))
    h, o = make_pair(capi, oracle_mod, prob, mode=mode)
    rng = np.random.default_rng(0)
    Q = prob.Q0.copy()
    oi = prob.desc.opt_index
    Q[:, oi, 2:] += 0.05 * rng.standard_normal(Q[:, oi, 2:].shape)
    a = h.eval_objective(0, prob.goals, n_goals, prob.S, prob.base, Q)
    b = o.eval_objective(0, prob.goals, n_goals, prob.S, prob.base, Q)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-10, atol=1e-14)  # f_goal (closed form in moments vs point sums)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12, atol=1e-14)  # f_obs
    np.testing.assert_allclose(a[2], b[2], rtol=1e-12, atol=1e-14)  # f_vel
    np.testing.assert_array_equal(a[3], b[3])                       # arg-min goal
    h.close()


@pytest.mark.parametrize("robot,dense", [("panda", True), ("fetch", True), ("panda", False), ("fetch", False)])
@pytest.mark.parametrize("mode", [0])
def test_obstacle_normal_equations_vs_oracle(capi, oracle_mod, robot, dense, mode):
    prob = Problem(robot, B=4, scene_seed=1, base=(0.0, 0.02, 0.01))
    h, o = make_pair(capi, oracle_mod, prob, mode=mode)
    Q = prob.Q0.copy()
    oi = prob.desc.opt_index
    if dense:
        # a dense random field: every surface point carries a cost value and a gradient
        rng = np.random.default_rng(5)
        sc = prob.scene
        ca = (0.05 * rng.random(sc.c_all.size)).astype(np.float32)
        co = (0.05 * rng.random(sc.c_all.size)).astype(np.float32)
        for s in (h, o):
            s.set_scene(0, ca, co, sc.shape, sc.origin, sc.res)
    else:
        # push the seeds into the table / objects of the synthetic scene
        Q[:, oi[1], 2:] += 1.0
        Q[:, oi] = np.clip(Q[:, oi], prob.desc.lower[oi][None, :, None], prob.desc.upper[oi][None, :, None])
    A, b, ss = h.eval_obstacle_normal_eq(0, prob.base, Q)
    Ao, bo, sso = o.eval_obstacle_normal_eq(0, prob.base, Q)
    assert np.abs(bo).max() > 1e-3 and sso.max() > 1e-4, "test problem exercises no obstacle terms"
    np.testing.assert_allclose(A[:, 2:], Ao[:, 2:], rtol=1e-9, atol=1e-11 * np.abs(Ao).max())
    np.testing.assert_allclose(b[:, 2:], bo[:, 2:], rtol=1e-9, atol=1e-11 * np.abs(bo).max())
    np.testing.assert_allclose(ss, sso, rtol=1e-12, atol=1e-16)
    h.close()


def test_plan_cost_vs_oracle(capi, oracle_mod):
    prob = Problem("panda", B=6, scene_seed=1)
    h, o = make_pair(capi, oracle_mod, prob)
    Q = prob.Q0.copy()
    Q[:, prob.desc.opt_index[1], :] += 1.0
    cg, dg = h.plan_cost(0, Q, prob.base[0])
    co, do = o.plan_cost(0, Q, prob.base[0])
    assert co.max() > 0
    np.testing.assert_allclose(cg, co, rtol=1e-12)
    np.testing.assert_allclose(dg, do, rtol=1e-14)
    # a values-only scene (gto_set_scene_values: no records, no distance fields) scores and looks up the same ...
    h.set_scene(3, prob.scene.c_all, prob.scene.c_obs, prob.scene.shape, prob.scene.origin, prob.scene.res, values_only=True)
    c3, d3 = h.plan_cost(3, Q, prob.base[0])
    np.testing.assert_array_equal(c3, cg)
    for a, b in zip(h.eval_points(3, prob.qc[:2], prob.base[:2]), h.eval_points(0, prob.qc[:2], prob.base[:2])):
        np.testing.assert_array_equal(a, b)
    # ... and every solve or objective entry point refuses it
    with pytest.raises(capi.GTOError, match="values-only"):
        h.solve_batch(3, prob.qc, prob.goals, prob.n_goals, prob.S, prob.base, prob.Q0)
    with pytest.raises(capi.GTOError, match="values-only"):
        h.eval_obstacle_normal_eq(3, prob.base, prob.Q0)
    h.close()


# ------------------------------------------------------------------------------------------ the solve
@pytest.mark.parametrize("mode", [0])
@pytest.mark.parametrize("max_iter", [0, 1, 2, 5])
def test_solver_iterates_match_oracle_step_by_step(capi, oracle_mod, max_iter, mode):
    prob = Problem("panda", B=5, scene_seed=3)
    h, o = make_pair(capi, oracle_mod, prob, mode=mode, max_iter=max_iter)
    Qg, dQg, fg, itg, stg = h.solve_batch(*prob.solve_args())
    Qo, dQo, fo, ito, sto = o.solve_batch(*prob.solve_args())
    np.testing.assert_array_equal(itg, ito)
    np.testing.assert_array_equal(stg, sto)
    np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-9)
    np.testing.assert_allclose(dQg, dQo, rtol=0, atol=1e-8)
    np.testing.assert_allclose(fg, fo, rtol=1e-9)
    h.close()


@pytest.mark.parametrize("robot,n_goals,standoff,grad_mode,scene_seed",
                         [("panda", 1, True, 0, 1), ("panda", 1, True, 0, 3), ("panda", 4, True, 0, 3),
                          ("panda", 1, False, 0, 2), ("panda", 1, True, 1, 3), ("fetch", 1, True, 0, 1)])
@pytest.mark.parametrize("mode", [0])
def test_full_solve_matches_oracle(capi, oracle_mod, robot, n_goals, standoff, grad_mode, scene_seed, mode):
    prob = Problem(robot, B=6, scene_seed=scene_seed, n_goals=n_goals, use_standoff=standoff)
    h, o = make_pair(capi, oracle_mod, prob, mode=mode, max_iter=60, grad_mode=grad_mode)
    Qg, dQg, fg, itg, stg = h.solve_batch(*prob.solve_args())
    Qo, dQo, fo, ito, sto = o.solve_batch(*prob.solve_args())
    np.testing.assert_array_equal(itg, ito)
    np.testing.assert_array_equal(stg, sto)
    np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-6)  # north_star tolerance
    np.testing.assert_allclose(fg, fo, rtol=1e-7)
    # structural invariants shared with the reference's stored plans
    d = prob.desc
    oi = d.opt_index
    assert np.abs(Qg[:, :, 1] - Qg[:, :, 0]).max() == 0.0
    np.testing.assert_array_equal(Qg[:, oi, 0], prob.qc[:, oi])
    assert (Qg[:, oi] >= d.lower[oi][None, :, None]).all() and (Qg[:, oi] <= d.upper[oi][None, :, None]).all()
    np.testing.assert_array_equal(Qg[:, d.param_index], prob.Q0[:, d.param_index])
    dt = 10.0 / 49
    np.testing.assert_allclose(Qg[:, :, :-1] + dt * dQg, Qg[:, :, 1:], atol=1e-15)  # Euler dynamics
    h.close()


@pytest.mark.parametrize("origin,n,res", [((0.15, -0.6, -0.1), 32, 0.04),      # the base and the first links are outside, low side of x
                                          ((-1.5, -1.4, -0.2), 32, 0.05),     # the arm reaches out of the grid on the high side of x
                                          ((0.2, -0.3, 0.05), 24, 0.03),      # a small grid in the middle of the workspace: most spheres outside
                                          ((-0.4, -1.12, 0.12), 48, 0.0467)])  # the table and the lower links are below the grid
@pytest.mark.parametrize("mode", [0])
def test_robot_partly_outside_the_grid_matches_oracle(capi, oracle_mod, origin, n, res, mode):
    """The broad phase culls a bounding sphere by the distance field at the CLIPPED voxel of its centre (round 3; before,
    spheres that stick out of the grid were never culled): robots that are partly or mostly outside the field, where the
    reference's lookups clip every index (gto/sdf_callback.py:90-114), must still match the oracle, which culls nothing."""
    prob = Problem("panda", B=6, scene_seed=3, n=n, res=res, n_goals=2, scene_origin=origin)
    h, o = make_pair(capi, oracle_mod, prob, mode=mode, max_iter=40)
    Qg, dQg, fg, itg, stg = h.solve_batch(*prob.solve_args())
    Qo, dQo, fo, ito, sto = o.solve_batch(*prob.solve_args())
    np.testing.assert_array_equal(itg, ito)
    np.testing.assert_array_equal(stg, sto)
    np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-6)
    np.testing.assert_allclose(fg, fo, rtol=1e-7)
    h.close()


@pytest.mark.parametrize("mode", [0])
def test_ragged_batches_and_scene_table(capi, oracle_mod, mode):
    """B not a multiple of 8, several scenes, per-instance goal counts, then the empty batch."""
    prob = Problem("panda", B=11, scene_seed=1, n_goals=3)
    h, o = make_pair(capi, oracle_mod, prob, mode=mode, max_iter=8)
    prob2 = Problem("panda", B=1, scene_seed=2)
    for s in (h, o):
        s.set_scene(9, prob2.scene.c_all, None, prob2.scene.shape, prob2.scene.origin, prob2.scene.res)
    sid = np.array([0, 9] * 5 + [0], dtype=np.int32)
    ng = np.array([1, 2, 3] * 3 + [3, 1], dtype=np.int32)
    args = (sid, prob.qc, prob.goals, ng, prob.S, prob.base, prob.Q0)
    Qg, _, fg, itg, stg = h.solve_batch(*args)
    Qo, _, fo, ito, sto = o.solve_batch(*args)
    np.testing.assert_array_equal(itg, ito)
    np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-7)
    np.testing.assert_allclose(fg, fo, rtol=1e-8)
    # empty batch is a no-op
    out = h.solve_batch(np.zeros(0, np.int32), np.zeros((0, 9)), np.zeros((0, 1, 16)), np.zeros(0, np.int32), None,
                        np.zeros((0, 3)), np.zeros((0, 9, 50)))
    assert out[0].shape == (0, 9, 50)
    # unknown scene / bad goal count are reported, not ignored
    with pytest.raises(capi.GTOError, match="scene"):
        h.solve_batch(np.full(11, 4, np.int32), *args[1:])
    with pytest.raises(capi.GTOError, match="n_goals"):
        h.solve_batch(sid, prob.qc, prob.goals, np.full(11, 7, np.int32), prob.S, prob.base, prob.Q0)
    h.drop_scene(9)
    with pytest.raises(capi.GTOError, match="scene"):
        h.solve_batch(*args)
    h.close()


@pytest.mark.parametrize("slots", [1, 5, 8, 64])
def test_slots_hand_over_to_waiting_instances(capi, oracle_mod, monkeypatch, slots):
    """A solve call keeps at most GTO_SLOTS instances in flight; an instance that finishes hands its slot to
    the next one that has not started.  Whatever the number of slots (fewer than, equal to, more than the batch),
    every instance gets bit-for-bit the trajectory it gets with all instances in flight from the start, and the
    oracle's iteration counts and status."""
    prob = Problem("panda", B=13, scene_seed=2, n_goals=2)
    h, o = make_pair(capi, oracle_mod, prob, max_iter=12)  # default: 512 slots, the whole batch in flight
    ref = h.solve_batch(*prob.solve_args())
    monkeypatch.setenv("GTO_SLOTS", str(slots))
    h2 = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], oracle_mod.reference_opts(max_iter=12), device=0)
    h2.set_mode(0)  # slots belong to the rounds mode, whatever GTO_MODE says
    h2.set_scene(*prob.scene_args())
    for _ in range(2):  # the second call reuses the workspace and the lists of the first
        got = h2.solve_batch(*prob.solve_args())
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
    Qo, _, fo, ito, sto = o.solve_batch(*prob.solve_args())
    np.testing.assert_array_equal(got[3], ito)
    np.testing.assert_array_equal(got[4], sto)
    np.testing.assert_allclose(got[0], Qo, rtol=0, atol=1e-7)
    assert len(set(ito.tolist())) > 1  # instances finish at different rounds: slots are handed over mid-solve
    h2.close()
    h.close()


@pytest.mark.parametrize("env", [{"GTO_PREBROAD": "0"}, {"GTO_SLOTS": "96"}, {"GTO_SLOTS": "96", "GTO_PREBROAD": "0"},
                                 {"GTO_OBS_TG": "2"}, {"GTO_OBS_TG": "5"}, {"GTO_FEW_INSTANCES": "16"}, {"GTO_PB_MIN_GAIN": "2"},
                                 {"GTO_SLOTS": "1", "GTO_FEW_INSTANCES": "0", "GTO_OBS_TG": "1"},
                                 {"GTO_ITEM_GRID": "0"}, {"GTO_ITEM_HINT": "8"}, {"GTO_ITEM_HINT": "40", "GTO_SLOTS": "96"},
                                 {"GTO_ITEM_HINT": "300", "GTO_OBS_TG": "2"},
                                 {"GTO_STATIC_POS": "0"}, {"GTO_STATIC_POS": "0", "GTO_SLOTS": "96"}, {"GTO_PB_MERGE": "1"}, {"GTO_PB_MERGE": "100", "GTO_SLOTS": "96"}])
def test_step_kernel_broad_phase_does_not_change_results(capi, oracle_mod, monkeypatch, env):
    """In the rounds that fill the GPU the step kernel tests the bounding spheres of its new trial trajectory itself
    (prebroad_tail: serial kinematics per lane instead of the obstacle kernel's matrix-core prefix), settles the waypoint
    groups none of whose spheres can reach a non-zero voxel record -- exact zeros either way -- and the obstacle launch is
    laid out over the groups that are left.  Switching that off, refilling positions mid-call (96 positions for 160
    instances; one position and one waypoint per group: the item list at its shortest against the launch's rounding),
    changing the group size, or switching it off mid-call (GTO_PB_MIN_GAIN=2: after round 12) gives bit-for-bit
    the same trajectories, costs and iteration counts; and they match the oracle, which culls nothing.  The itemized launch
    is laid out over an estimate of the list's length with a crew of 64 looping workgroups behind it for the rest
    (GTO_ITEM_GRID=0: over the upper bound, no crew; GTO_ITEM_HINT: the estimate itself -- 8: the crew does nearly
    everything, several items per workgroup).  GTO_STATIC_POS=0: every instance draws its positions in the next round's lists
    from the lists' counters instead of keeping them (with refills: 96 positions); GTO_PB_MERGE: chunks of a link under one
    sphere of the step kernel's test (1: the chunks' own spheres, 100: whole links)."""
    prob = Problem("panda_5k", B=160, scene_seed=5, n=64, res=0.035, n_goals=1)
    # (a call of 160 would run in the launches for few instances from its first round: the broad phase belongs to the others)
    monkeypatch.setenv("GTO_FEW_INSTANCES", "64")
    h, o = make_pair(capi, oracle_mod, prob, max_iter=40)
    h.set_profiling(True)
    ref = h.solve_batch(*prob.solve_args())
    prof = h.last_kernel_profile()
    h.set_profiling(False)
    assert prof["k_lm_step<4,1>"][1] > 10, prof  # launches of the step kernel variant that carries the broad phase: the path under test ran
    ref = h.solve_batch(*prob.solve_args())
    for kn, va in env.items():
        monkeypatch.setenv(kn, va)
    h2 = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], oracle_mod.reference_opts(max_iter=40), device=0)
    h2.set_mode(0)
    h2.set_scene(*prob.scene_args())
    got = h2.solve_batch(*prob.solve_args())
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)
    sub = slice(0, 12)
    args = list(prob.solve_args())
    args_sub = (args[0], args[1][sub], args[2][sub], args[3], args[4], args[5][sub], args[6][sub])
    Qo, dQo, fo, ito, sto = o.solve_batch(*args_sub)
    np.testing.assert_array_equal(ref[3][sub], ito)
    np.testing.assert_allclose(ref[0][sub], Qo, rtol=0, atol=1e-6)
    h2.close()
    h.close()


@pytest.mark.parametrize("robot,shelf,T", [("fetch", False, 50), ("fetch", True, 50), ("fetch", False, 4), ("panda", False, 5), ("panda_5k", False, 30)])
def test_step_kernel_broad_phase_other_robots_and_horizons(capi, oracle_mod, monkeypatch, robot, shelf, T):
    """The same on a robot with another link count and collision links on fixed frames (static links: never tested, never
    settled) and, in the shelf, next to nothing to settle (the call switches the broad phase of the step kernel off after
    round 12); at horizons down to T = 4, where the step kernel's dead LDS holds the visual transforms of a waypoint or two
    per pass: bit-identical with and without it, and equal to the oracle."""
    so = -10 if T >= 30 else -1
    prob = Problem(robot, B=130, scene_seed=4, n=64, res=0.035, n_goals=1, shelf=shelf, T=T)
    opts = lambda: oracle_mod.reference_opts(max_iter=30, T=T, standoff_offset=so)
    monkeypatch.setenv("GTO_FEW_INSTANCES", "64")  # (130 instances: otherwise few-instance launches throughout, no broad phase in the step kernel)
    h, o = make_pair(capi, oracle_mod, prob, max_iter=30, T=T, standoff_offset=so)
    ref = h.solve_batch(*prob.solve_args())
    monkeypatch.setenv("GTO_PREBROAD", "0")
    h2 = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts(), device=0)
    h2.set_mode(0)
    h2.set_scene(*prob.scene_args())
    got = h2.solve_batch(*prob.solve_args())
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)
    sub = slice(0, 10)
    args = list(prob.solve_args())
    Qo, dQo, fo, ito, sto = o.solve_batch(args[0], args[1][sub], args[2][sub], args[3], args[4], args[5][sub], args[6][sub])
    np.testing.assert_array_equal(ref[3][sub], ito)
    np.testing.assert_allclose(ref[0][sub], Qo, rtol=0, atol=1e-6)
    h2.close()
    h.close()


@pytest.mark.parametrize("knob,value", [("GTO_OBS_TG", "1"), ("GTO_OBS_TG", "2"), ("GTO_OBS_TG", "4"), ("GTO_OBS_INTERLEAVE", "0"),
                                        ("GTO_OBS_INTERLEAVE", "1"), ("GTO_AHEAD", "1"), ("GTO_AHEAD", "24"),
                                        ("GTO_STEP_NW_FEW", "4"), ("GTO_DIST_RELAX", "1"), ("GTO_FEW_INSTANCES", "0"),
                                        ("GTO_FEW_INSTANCES", "8"), ("GTO_SPEC_REJ", "1"), ("GTO_SPEC_REJ", "2"), ("GTO_SPEC_REJ", "3"),
                                        ("GTO_SPEC_ACC,GTO_SPEC_DEEP,GTO_SPEC_JOBS,GTO_SPEC_STREAK", "4,100000,100000,0"),
                                        ("GTO_SPEC_ACC,GTO_SPEC_DEEP,GTO_SPEC_REJ,GTO_SPEC_JOBS,GTO_SPEC_STREAK", "2,100000,1,100000,0"),
                                        ("GTO_SPEC_ACC,GTO_SPEC_DEEP,GTO_SPEC_REJ", "1,0,1"), ("GTO_OBS_DEEP", "0"),
                                        ("GTO_SPEC_FEW,GTO_SPEC_REJ,GTO_SPEC_JOBS", "100000,4,100000"), ("GTO_SPEC_FEW", "0"),
                                        ("GTO_SPEC_STREAK,GTO_SPEC_JOBS", "1,100000"), ("GTO_SPEC_STREAK,GTO_SPEC_JOBS", "0,100000"),
                                        ("GTO_SPEC_JOBS,GTO_SPEC_REJ_FEW", "1,1"), ("GTO_SPEC_JOBS,GTO_SPEC_REJ_FEW", "48,4")])
def test_launch_geometry_does_not_change_results(capi, oracle_mod, monkeypatch, knob, value):
    """How the waypoints are dealt to the workgroups of the obstacle kernel (group size, consecutive or interleaved), how
    far the host runs ahead of the GPU, when a call switches to the launches for few instances in flight, and how many
    candidate trial points a step hands out ahead of their evaluation (speculation: after a rejection the next trial point
    is known; GTO_SPEC_*: per launch by the number of instances in flight, per instance by its run of first-try accepts)
    are scheduling decisions: every instance gets bit-for-bit the same trajectory, cost and iteration
    count (a (waypoint, link) key is folded by one wave in chunk order, the keys of a waypoint are added up in link order;
    the candidates of a step are walked in the order the sequential algorithm would have met them)."""
    prob = Problem("panda", B=24, scene_seed=5, n_goals=2)
    h, o = make_pair(capi, oracle_mod, prob, max_iter=15)
    ref = h.solve_batch(*prob.solve_args())
    for kn, va in zip(knob.split(","), value.split(",")):
        monkeypatch.setenv(kn, va)
    h2 = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], oracle_mod.reference_opts(max_iter=15), device=0)
    h2.set_mode(0)
    h2.set_scene(*prob.scene_args())
    for _ in range(2):
        got = h2.solve_batch(*prob.solve_args())
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
    if knob == "GTO_DIST_RELAX":
        # the distance field built by 48 relaxation sweeps (h2) and by one separable pass per axis (h) cull the same
        # chunks: the obstacle kernel gathers exactly the same number of surface points
        work = []
        for x in (h, h2):
            x.set_profiling(True)
            x.solve_batch(*prob.solve_args())
            work.append(x.last_kernel_work()[0])
        assert work[0] == work[1] and work[0] > 0
    h2.close()
    h.close()


@pytest.mark.parametrize("robot,B,T", [("panda", 96, 50), ("fetch", 40, 30), ("fetch_mobile", 24, 20)])
def test_lanes_of_a_call_do_not_change_results(capi, oracle_mod, robot, B, T):
    """gto_set_lanes: the instances of a call dealt to several lanes (streams and lists of their own over one workspace,
    one host thread), and the lanes' last instances handed to lane 0 (k_adopt), are scheduling: every instance gets
    bit-for-bit the trajectory, cost, iteration count and status of the call that runs as one lane.  Covered: uneven
    ranges, more lanes than the GPU has hardware queues, hand-over of everything at once (adopt_below >= a lane's share)
    and of the last stragglers only, no hand-over, the wide step kernel (position-indexed scratch per lane), a second call
    on the same handle with other lane settings (buffers are re-dealt)."""
    prob = Problem(robot, B=B, scene_seed=3, T=T, n_goals=2 if robot == "panda" else 1)
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-max(2, T // 5), max_iter=40)
    h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    prob.finish(h.eval_fk)
    h.set_scene(*prob.scene_args())
    h.set_lanes(1, 1, 0)
    ref = h.solve_batch(*prob.solve_args())
    assert len(set(ref[3].tolist())) > 2  # iteration counts differ: lanes empty at different times
    for lanes, per_lane, adopt in ((4, 8, 0), (4, 8, 6), (3, 7, 1000), (8, 1, 3), (2, B // 2, 2), (4, 8, 1)):
        h.set_lanes(lanes, per_lane, adopt)
        for _ in range(2):
            got = h.solve_batch(*prob.solve_args())
            for a, b in zip(ref, got):
                np.testing.assert_array_equal(a, b)
    h.close()


@pytest.mark.parametrize("mode", [0])
@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_hip_lm_ends_where_lbfgsb_ends(capi, oracle_mod, robot, mode):
    """The solver iteration against a third-party optimiser (tests/independent.py): on the smooth problem IPOPT sees
    gradient-wise (GTO_GRAD_ZERO, empty field), every HIP solution is a KKT point, and where SciPy's L-BFGS-B ends in
    the same basin it ends at the same trajectory and the same objective value."""
    from independent import check_against_lbfgsb
    T = 12
    prob = Problem(robot, B=4, scene_seed=1, T=T)
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-3, grad_mode=1, max_iter=300, tol_rel_f=1e-14)
    h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    h.set_mode(mode)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    prob.finish(h.eval_fk)
    zero = np.zeros_like(prob.scene.c_all)
    for x in (h, o):
        x.set_scene(0, zero, zero, prob.scene.shape, prob.scene.origin, prob.scene.res)
    Q, _, f, it, st = h.solve_batch(*prob.solve_args())
    check_against_lbfgsb(o, prob, opts, Q, f, st, min_same_basin=2)
    h.close()


def test_hip_lm_ends_where_lbfgsb_ends_full_horizon(capi, oracle_mod):
    """The same at the reference's horizon (T = 50, standoff ten waypoints from the end; 336 free variables) on a field that
    is not empty: a sparse one whose non-zero voxels (the outermost layers of the grid) the arm never reaches, so that the
    obstacle kernel runs its broad phase against a real distance field while the objective stays the smooth one
    (GTO_GRAD_ZERO).  Every HIP solution is a KKT point; where L-BFGS-B ends in the same basin, same trajectory, same minimum."""
    from independent import check_against_lbfgsb
    T = 50
    prob = Problem("panda", B=3, scene_seed=1, T=T)
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-10, grad_mode=1, max_iter=600, tol_rel_f=1e-14)
    h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    h.set_mode(0)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    prob.finish(h.eval_fk)
    c = np.zeros(prob.scene.shape, dtype=np.float32)
    c[-3:, :, :] = 0.01
    for x in (h, o):
        x.set_scene(0, c.reshape(-1), c.reshape(-1), prob.scene.shape, prob.scene.origin, prob.scene.res)
    Q, _, f, it, st = h.solve_batch(*prob.solve_args())
    Qo, _, fo, ito, sto = o.solve_batch(*prob.solve_args())
    np.testing.assert_array_equal(it, ito)
    np.testing.assert_allclose(Q, Qo, rtol=0, atol=1e-6)
    check_against_lbfgsb(o, prob, opts, Q, f, st, min_same_basin=2)
    h.close()


@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_hip_ik_ends_where_lbfgsb_ends(capi, oracle_mod, robot):
    """gto_solve_ik_batch (no collision term) against SciPy's L-BFGS-B on the objective value (tests/independent.py)."""
    from independent import check_ik_against_lbfgsb
    prob = Problem(robot, B=8, scene_seed=5)
    h, o = make_pair(capi, oracle_mod, prob, tol_rel_f=1e-14)
    rng = np.random.default_rng(1)
    oi = prob.desc.opt_index
    q0 = prob.qc.copy()
    q0[4:, oi] = prob.qgoal[4:, 0][:, oi] + rng.uniform(-0.3, 0.3, size=(4, len(oi)))
    q, f, it, st = h.solve_ik_batch(None, q0, prob.goals[:, 0], prob.base, max_iter=200)
    check_ik_against_lbfgsb(o, prob, q0, q, f, min_agree=6)
    h.close()


def test_hip_base_placement_ends_where_lbfgsb_ends(capi, oracle_mod):
    """gto_solve_base_batch with a firm effort weight against SciPy's L-BFGS-B on the objective value."""
    from independent import check_base_against_lbfgsb
    from grasptrajopt_amd import synthetic as syn
    prob = Problem("fetch", B=4, scene_seed=2)
    h, o = make_pair(capi, oracle_mod, prob, tol_rel_f=1e-14)
    goals, _ = syn.make_base_goal_sets(prob.desc, h.eval_fk, prob.cfg["link_ee"], prob.qc[0], 4, 3, 0)
    y, q, f, it, st = h.solve_base_batch(prob.qc, goals, None, 1.0, max_iter=300)
    check_base_against_lbfgsb(o, prob.desc, prob.qc, np.asarray(goals).reshape(4, 3, 16), 1.0, y, q, f, min_agree=3)
    h.close()


@pytest.mark.parametrize("mode", [0])
@pytest.mark.parametrize("robot", ["panda", "fetch", "panda_5k"])
def test_hip_obstacle_blocks_against_finite_difference_jacobians(capi, oracle_mod, robot, mode):
    """k_obstacle_gram's blocks (wrench Grams per link on the matrix core, projected onto the joint screws) against a
    point-by-point numpy assembly with FINITE-DIFFERENCE point Jacobians (tests/independent.py): no analytic Jacobian
    of anybody's is trusted."""
    from independent import check_obstacle_blocks_against_fd
    T = 16
    prob = Problem(robot, B=2, scene_seed=3, T=T)
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-4)
    h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    h.set_mode(mode)
    prob.finish(h.eval_fk)
    h.set_scene(*prob.scene_args())
    nz = sum(check_obstacle_blocks_against_fd(h, prob.desc, T, T - 4, prob.Q0[b], prob.base[b]) for b in range(prob.B))
    assert nz >= 4
    h.close()


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_robots_match_oracle(capi, oracle_mod, seed):
    """Random kinematic trees (tests/helpers.random_robot: branching, prismatic and fixed joints anywhere, 3-8 optimised
    joints, side branches with parameter joints, up to 24 frames): forward kinematics of every frame, objective terms,
    obstacle normal equations and the solve itself, HIP against the oracle."""
    from helpers import random_robot
    desc, ee = random_robot(seed)
    T, B = 20, 6
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-4, max_iter=25)
    h = capi.SolverHandle(desc, ee, ee, opts, device=0, n_gripper_points=40)
    o = oracle_mod.Oracle(desc, ee, ee, opts, n_gripper_points=40)
    rng = np.random.default_rng(100 + seed)
    lo, hi = desc.lower, desc.upper
    q = rng.uniform(lo, hi, size=(16, desc.ndof))
    np.testing.assert_allclose(h.eval_fk(q), o.eval_fk(q), rtol=0, atol=1e-12)
    # a field around the robot: a dense random band so that every link meets cost and gradient somewhere
    n, res = 40, 0.05
    origin = (-1.0, -1.0, -1.0)
    c_all = (0.03 * rng.random(n ** 3) * (rng.random(n ** 3) < 0.3)).astype(np.float32)
    c_obs = (0.03 * rng.random(n ** 3) * (rng.random(n ** 3) < 0.2)).astype(np.float32)
    for x in (h, o):
        x.set_scene(0, c_all, c_obs, (n, n, n), origin, res)
    qc = rng.uniform(0.3 * lo, 0.3 * hi, size=(B, desc.ndof))
    qg = rng.uniform(0.8 * lo, 0.8 * hi, size=(B, desc.ndof))
    qg[:, desc.param_index] = qc[:, desc.param_index]
    fe = desc.frame_index(ee)
    goals = o.eval_fk(qg)[:, fe].reshape(B, 1, 16)
    S = syn.standoff_pose(-0.05, "z")
    base = np.zeros((B, 3))
    Q0 = np.stack([syn.make_seed(qc[b], qg[b], T, desc.param_index) for b in range(B)])
    a = h.eval_objective(0, goals, 1, S, base, Q0)
    b_ = o.eval_objective(0, goals, 1, S, base, Q0)
    for x, y in zip(a[:3], b_[:3]):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-13)
    A, g, ss = h.eval_obstacle_normal_eq(0, base, Q0)
    Ao, go, sso = o.eval_obstacle_normal_eq(0, base, Q0)
    np.testing.assert_allclose(A[:, 2:], Ao[:, 2:], rtol=1e-8, atol=1e-10 * max(np.abs(Ao).max(), 1e-30))
    np.testing.assert_allclose(g[:, 2:], go[:, 2:], rtol=1e-8, atol=1e-10 * max(np.abs(go).max(), 1e-30))
    np.testing.assert_allclose(ss, sso, rtol=1e-11, atol=1e-15)
    for mode in (0,):
        h.set_mode(mode)
        Qg, _, fg, itg, stg = h.solve_batch(0, qc, goals, 1, S, base, Q0)
        Qo, _, fo, ito, sto = o.solve_batch(0, qc, goals, 1, S, base, Q0)
        np.testing.assert_array_equal(itg, ito)
        np.testing.assert_array_equal(stg, sto)
        np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-6)
        np.testing.assert_allclose(fg, fo, rtol=1e-8)
    # inverse kinematics of the same goals (whole LM loop in one workgroup), without and with the collision term
    for sid in (None, 0):
        qi, fi, iti, sti = h.solve_ik_batch(sid, qc, goals[:, 0], base, max_iter=40)
        qo2, fo2, ito2, sto2 = o.solve_ik_batch(sid, qc, goals[:, 0], base, max_iter=40)
        np.testing.assert_array_equal(iti, ito2)
        np.testing.assert_array_equal(sti, sto2)
        np.testing.assert_allclose(qi, qo2, rtol=0, atol=1e-6)
    h.close()


@pytest.mark.parametrize("seed,n_opt", [(50, 9), (51, 11), (52, 12), (53, 14), (54, 16), (55, 10)])
def test_random_wide_robots_match_oracle(capi, oracle_mod, seed, n_opt):
    """Nine to sixteen optimised joints on random trees (16-wide blocks, k_lm_step_wide): the solve against the oracle."""
    from helpers import random_robot
    desc, ee = random_robot(seed, n_frames=max(n_opt + 4, 14), n_opt=n_opt)
    assert desc.n_opt == n_opt
    T, B = 16, 4
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-3, max_iter=20)
    h = capi.SolverHandle(desc, ee, ee, opts, device=0, n_gripper_points=40)
    o = oracle_mod.Oracle(desc, ee, ee, opts, n_gripper_points=40)
    rng = np.random.default_rng(200 + seed)
    lo, hi = desc.lower, desc.upper
    n, res = 40, 0.06
    c_all = (0.03 * rng.random(n ** 3) * (rng.random(n ** 3) < 0.3)).astype(np.float32)
    c_obs = (0.03 * rng.random(n ** 3) * (rng.random(n ** 3) < 0.2)).astype(np.float32)
    for x in (h, o):
        x.set_scene(0, c_all, c_obs, (n, n, n), (-1.2, -1.2, -1.2), res)
    qc = rng.uniform(0.3 * lo, 0.3 * hi, size=(B, desc.ndof))
    qg = rng.uniform(0.8 * lo, 0.8 * hi, size=(B, desc.ndof))
    qg[:, desc.param_index] = qc[:, desc.param_index]
    goals = o.eval_fk(qg)[:, desc.frame_index(ee)].reshape(B, 1, 16)
    S = syn.standoff_pose(-0.05, "z")
    base = np.zeros((B, 3))
    Q0 = np.stack([syn.make_seed(qc[b], qg[b], T, desc.param_index) for b in range(B)])
    a = h.eval_objective(0, goals, 1, S, base, Q0)
    b_ = o.eval_objective(0, goals, 1, S, base, Q0)
    for x, y in zip(a[:3], b_[:3]):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-13)
    Qg, _, fg, itg, stg = h.solve_batch(0, qc, goals, 1, S, base, Q0)
    Qo, _, fo, ito, sto = o.solve_batch(0, qc, goals, 1, S, base, Q0)
    np.testing.assert_array_equal(itg, ito)
    np.testing.assert_array_equal(stg, sto)
    np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-6)
    np.testing.assert_allclose(fg, fo, rtol=1e-8)
    h.close()


@pytest.mark.parametrize("seed", list(range(60, 70)))
def test_random_edge_cases_match_oracle(capi, oracle_mod, seed):
    """Edge cases the benchmark workloads never meet, on random trees: links with 1, 64, 65 or several hundred surface
    points (chunk boundaries), tiny and anisotropic grids, most of the robot outside the grid (clipped voxel indices),
    a base offset, ragged goal sets of up to nine goals, standoff on and off, short horizons."""
    from helpers import random_robot
    rng = np.random.default_rng(300 + seed)
    desc, ee = random_robot(seed)
    counts = rng.choice([1, 2, 63, 64, 65, 128, 129, 300], size=desc.n_links)
    pts, plink = [], []
    for l, m in enumerate(counts):
        pts.append(rng.uniform(-0.04, 0.04, 3) + rng.standard_normal((m, 3)) * 0.03)
        plink.append(np.full(m, l, dtype=np.int32))
    desc.points, desc.point_link = np.concatenate(pts), np.concatenate(plink)
    desc.normals = np.zeros_like(desc.points)
    T = int(rng.choice([4, 5, 9, 20]))
    off = -int(rng.integers(1, max(2, T - 2)))
    use_so = bool(rng.integers(0, 2))
    B, n_max = 5, 9
    opts = oracle_mod.reference_opts(T=T, standoff_offset=off, max_iter=15, grad_mode=int(rng.integers(0, 2)))
    h = capi.SolverHandle(desc, ee, ee, opts, device=0, n_gripper_points=30)
    o = oracle_mod.Oracle(desc, ee, ee, opts, n_gripper_points=30)
    shape = tuple(int(v) for v in rng.choice([2, 3, 5, 8, 17, 33], size=3))
    res = float(rng.choice([0.02, 0.07, 0.3]))
    origin = tuple(rng.uniform(-0.6, 0.1, 3))
    nv = int(np.prod(shape))
    c_all = (0.05 * rng.random(nv) * (rng.random(nv) < 0.6)).astype(np.float32)
    c_obs = (0.05 * rng.random(nv) * (rng.random(nv) < 0.4)).astype(np.float32)
    for x in (h, o):
        x.set_scene(0, c_all, c_obs, shape, origin, res)
    lo, hi = desc.lower, desc.upper
    qc = rng.uniform(0.3 * lo, 0.3 * hi, size=(B, desc.ndof))
    n_goals = rng.integers(1, n_max + 1, size=B).astype(np.int32)
    qg = rng.uniform(0.8 * lo, 0.8 * hi, size=(B, n_max, desc.ndof))
    goals = o.eval_fk(qg.reshape(-1, desc.ndof))[:, desc.frame_index(ee)].reshape(B, n_max, 16)
    S = syn.standoff_pose(-0.05, "z") if use_so else None
    base = rng.uniform(-0.2, 0.2, size=(B, 3))
    Q0 = np.stack([syn.make_seed(qc[b], np.where(np.isin(np.arange(desc.ndof), desc.param_index), qc[b], qg[b, 0]), T, desc.param_index) for b in range(B)])
    a = h.eval_objective(0, goals, n_goals, S, base, Q0)
    b_ = o.eval_objective(0, goals, n_goals, S, base, Q0)
    for x, y in zip(a[:3], b_[:3]):
        np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(a[3], b_[3])
    xyz, offs, val, grad = h.eval_points(0, qc, base, use_obs=True)
    xo, oo, vo, go = o.eval_points(0, qc, base, use_obs=True)
    np.testing.assert_array_equal(offs, oo)          # clipped voxel indices, bit for bit
    np.testing.assert_array_equal(val, vo)
    for mode in (0,):
        h.set_mode(mode)
        Qg, _, fg, itg, stg = h.solve_batch(0, qc, goals, n_goals, S, base, Q0)
        Qo, _, fo, ito, sto = o.solve_batch(0, qc, goals, n_goals, S, base, Q0)
        np.testing.assert_array_equal(itg, ito)
        np.testing.assert_array_equal(stg, sto)
        np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-6)
    h.close()


@pytest.mark.parametrize("seed", list(range(80, 88)))
def test_random_robots_base_placement_matches_oracle(capi, oracle_mod, seed):
    """Base placement (row f-4) on random trees: ragged goal sets, with and without the effort term."""
    from helpers import random_robot
    desc, ee = random_robot(seed)
    rng = np.random.default_rng(400 + seed)
    h = capi.SolverHandle(desc, ee, ee, oracle_mod.reference_opts(), device=0, n_gripper_points=30)
    o = oracle_mod.Oracle(desc, ee, ee, oracle_mod.reference_opts(), n_gripper_points=30)
    B, n_max = 5, 6
    qc = rng.uniform(0.3 * desc.lower, 0.3 * desc.upper, size=(B, desc.ndof))
    goals = np.zeros((B, n_max, 4, 4))
    for b in range(B):
        gb, _ = syn.make_base_goal_sets(desc, o.eval_fk, ee, qc[b], 1, n_max, seed * 10 + b, spread=0.4, shift=0.2, turn=0.4)
        goals[b] = gb[0]
    n_goals = rng.integers(1, n_max + 1, size=B).astype(np.int32)
    for w, iters in ((0.0, 60), (0.5, 25)):
        yg, qg, fg, itg, stg = h.solve_base_batch(qc, goals, n_goals, w, max_iter=iters)
        yo, qo, fo, ito, sto = o.solve_base_batch(qc, goals, n_goals, w, max_iter=iters)
        np.testing.assert_array_equal(itg, ito)
        np.testing.assert_array_equal(stg, sto)
        np.testing.assert_allclose(fg, fo, rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(yg, yo, rtol=0, atol=1e-6)
    h.close()


def test_full_size_properties(capi, oracle_mod):
    """BASELINE.json configs[1] sizes (Panda, ~5k surface points, 128^3 field, 64 goals, T=50):
    size-independent properties on all 64 instances, and eight of them against the oracle at the reference's full
    iteration cap (100), the oracle running on every core the box allows."""
    prob = Problem("panda_5k", B=64, scene_seed=3, n=128, res=0.0175)
    opts = oracle_mod.reference_opts(max_iter=100)
    h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    prob.finish(h.eval_fk)
    h.set_scene(*prob.scene_args())
    Q, dQ, f, it, st = h.solve_batch(*prob.solve_args())
    d = prob.desc
    oi = d.opt_index
    assert (Q[:, oi] >= d.lower[oi][None, :, None]).all() and (Q[:, oi] <= d.upper[oi][None, :, None]).all()
    assert np.abs(Q[:, :, 1] - Q[:, :, 0]).max() == 0.0
    np.testing.assert_array_equal(Q[:, oi, 0], prob.qc[:, oi])
    # reported cost == objective re-evaluated at the returned trajectory; never above the seed's
    fg, fo, fv, _ = h.eval_objective(0, prob.goals, 1, prob.S, prob.base, Q)
    np.testing.assert_allclose(fg + fo + fv, f, rtol=1e-10)
    seed = prob.Q0.copy()
    seed[:, :, :2] = prob.qc[:, :, None]
    seed[:, oi] = np.clip(seed[:, oi], d.lower[oi][None, :, None], d.upper[oi][None, :, None])
    sg, so, sv, _ = h.eval_objective(0, prob.goals, 1, prob.S, prob.base, seed)
    assert (f <= (sg + so + sv) * (1 + 1e-12)).all()
    # restarting from the solution never raises the objective
    Q2, _, f2, it2, _ = h.solve_batch(0, prob.qc, prob.goals, 1, prob.S, prob.base, Q)
    assert (f2 <= f * (1 + 1e-9)).all()
    # a sample of instances against the oracle (same algorithm, FP64)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    o.set_scene(*prob.scene_args())
    sel = [0, 9, 17, 26, 35, 44, 53, 63]
    Qo, _, fo_, ito, sto = o.solve_batch(0, prob.qc[sel], prob.goals[sel], 1, prob.S, prob.base[sel], prob.Q0[sel], n_threads=o.usable_cores())
    np.testing.assert_array_equal(it[sel], ito)
    np.testing.assert_array_equal(st[sel], sto)
    np.testing.assert_allclose(Q[sel], Qo, rtol=0, atol=1e-6)
    np.testing.assert_allclose(f[sel], fo_, rtol=1e-9)
    h.close()


def test_pipelined_batches_equal_serial_solves(capi, oracle_mod):
    """Three handles, each bound to its own stream (gto_set_stream), driven concurrently by
    BatchPipeline: every batch must come back exactly as one handle solving them one by one."""
    import torch
    from grasptrajopt_amd.parallel import BatchPipeline
    probs = [Problem("panda", B=b, scene_seed=s) for b, s in ((9, 1), (16, 2), (5, 3), (12, 4), (7, 5), (16, 6))]
    opts = oracle_mod.reference_opts(max_iter=12)
    streams = [torch.cuda.Stream(torch.device("cuda", 0)) for _ in range(3)]
    hs = []
    for st in streams:
        h = capi.SolverHandle(probs[0].desc, probs[0].cfg["link_ee"], probs[0].cfg["link_gripper"], opts, device=0)
        h.set_stream(st.cuda_stream)
        hs.append(h)
    for i, p in enumerate(probs):
        p.finish(hs[0].eval_fk)
        hs[0].set_scene(i, p.scene.c_all, p.scene.c_obs, p.scene.shape, p.scene.origin, p.scene.res)
        hs[1].share_scene(i, hs[0])  # borrowed: no second copy in HBM
        hs[2].set_scene(i, p.scene.c_all, p.scene.c_obs, p.scene.shape, p.scene.origin, p.scene.res)
    batches = [(np.full(p.B, i, np.int32), p.qc, p.goals, 1, p.S, p.base, p.Q0) for i, p in enumerate(probs)]
    serial = [hs[0].solve_batch(*b) for b in batches]
    with BatchPipeline(hs) as pipe:
        piped = pipe.solve_batches(batches * 3)
    for k, out in enumerate(piped):
        ref = serial[k % len(batches)]
        np.testing.assert_array_equal(out[3], ref[3])
        np.testing.assert_array_equal(out[0], ref[0])  # bit-identical trajectories
        np.testing.assert_array_equal(out[2], ref[2])
    # back to a private stream
    hs[1].set_stream(None)
    again = hs[1].solve_batch(*batches[0])
    np.testing.assert_array_equal(again[0], serial[0][0])
    # a borrowed scene can be dropped without touching the owner's copy; sharing an unknown scene fails
    hs[1].drop_scene(0)
    np.testing.assert_array_equal(hs[0].solve_batch(*batches[0])[0], serial[0][0])
    with pytest.raises(capi.GTOError, match="scene"):
        hs[1].share_scene(0, hs[0], 77)
    for h in (hs[1], hs[2], hs[0]):  # the owner of the shared scenes goes last
        h.close()


@pytest.mark.parametrize("T,offset", [(4, -1), (5, -2), (30, -10), (64, -10), (80, -10), (96, -20)])
def test_other_horizons_match_oracle(capi, oracle_mod, T, offset):
    """The reference fixes T = 50 (gto/gto_planner.py:25); BASELINE.json also names 30- and 80-waypoint
    variants.  Shortest legal horizon, the old 64 limit, and the current maximum (96)."""
    prob = Problem("panda", B=3, scene_seed=4, T=T)
    h, o = make_pair(capi, oracle_mod, prob, T=T, standoff_offset=offset, max_iter=10)
    Qg, dQg, fg, itg, stg = h.solve_batch(*prob.solve_args())
    Qo, dQo, fo, ito, sto = o.solve_batch(*prob.solve_args())
    np.testing.assert_array_equal(itg, ito)
    np.testing.assert_array_equal(stg, sto)
    np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-6)
    np.testing.assert_allclose(fg, fo, rtol=1e-8)
    h.close()
    with pytest.raises(capi.GTOError, match="T must be"):
        capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"],
                          oracle_mod.reference_opts(T=97, standoff_offset=-10), device=0)


@pytest.mark.parametrize("robot,collide,grad_mode", [("panda", False, 0), ("panda", True, 0), ("panda", True, 1),
                                                      ("fetch", True, 0), ("fetch", False, 0)])
def test_ik_matches_oracle(capi, oracle_mod, robot, collide, grad_mode):
    """gto_solve_ik_batch (one workgroup per goal, whole LM loop on the GPU) against the CPU restatement of
    gto/ik_solver.py: same iteration counts and status, q within 1e-6 rad, objective to 1e-8 relative."""
    prob = Problem(robot, B=12, scene_seed=5)
    h, o = make_pair(capi, oracle_mod, prob, grad_mode=grad_mode)
    rng = np.random.default_rng(1)
    q0 = np.tile(np.array(prob.cfg["default_pose"]), (12, 1))
    oi = prob.desc.opt_index
    q0[6:, oi] = prob.qgoal[6:, 0][:, oi] + rng.uniform(-0.3, 0.3, size=(6, len(oi)))  # half far, half near seeds
    sid = 0 if collide else None
    qg, fg, itg, stg = h.solve_ik_batch(sid, q0, prob.goals[:, 0], prob.base, max_iter=50)
    qo, fo, ito, sto = o.solve_ik_batch(sid, q0, prob.goals[:, 0], prob.base, max_iter=50)
    np.testing.assert_array_equal(itg, ito)
    np.testing.assert_array_equal(stg, sto)
    np.testing.assert_allclose(qg, qo, rtol=0, atol=1e-6)
    np.testing.assert_allclose(fg, fo, rtol=1e-8, atol=1e-12)
    # max_iter = 0 returns the clipped seed
    q1, _, it1, st1 = h.solve_ik_batch(sid, q0, prob.goals[:, 0], prob.base, max_iter=0)
    assert (it1 == 0).all() and (st1 == 1).all()
    np.testing.assert_allclose(q1[:, oi], np.clip(q0[:, oi], prob.desc.lower[oi], prob.desc.upper[oi]), atol=0)
    with pytest.raises(capi.GTOError, match="scene"):
        h.solve_ik_batch(7, q0, prob.goals[:, 0], prob.base)
    h.close()


@pytest.mark.parametrize("robot,n", [("fetch", 10), ("panda", 32), ("fetch", 1)])
def test_base_placement_matches_oracle(capi, oracle_mod, robot, n):
    """gto_solve_base_batch (one workgroup per goal set, arrow system eliminated on the GPU) against the CPU
    restatement of gto/base_planner.py: same iteration counts and status; base pose and joint angles within
    1e-6 (m, rad) where the problem is well conditioned (no effort term, or a bounded number of iterations
    with it); with the tiny effort weight run to the cap the minimiser sits in a nearly flat valley (the arm
    can absorb base motion), so there the objective is compared (1e-6 relative) and the point only coarsely.
    Ragged goal sets and the padding rows of q_out are covered as well."""
    from grasptrajopt_amd import load_builtin, synthetic as syn
    cfg, desc = cfg_of(robot), load_builtin(robot)
    opts = oracle_mod.reference_opts()
    h = capi.SolverHandle(desc, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
    o = oracle_mod.Oracle(desc, cfg["link_ee"], cfg["link_gripper"], opts, n_gripper_points=100)
    qc = np.array(cfg["default_pose"], dtype=np.float64)
    B = 9
    goals, _ = syn.make_base_goal_sets(desc, h.eval_fk, cfg["link_ee"], qc, B, n, seed=3)
    QC = np.tile(qc, (B, 1))
    rng = np.random.default_rng(2)
    QC[1:, desc.opt_index] += rng.uniform(-0.1, 0.1, size=(B - 1, len(desc.opt_index)))  # per-set current configuration
    ng = rng.integers(1, n + 1, size=B).astype(np.int32)
    ng[0] = n
    for w, cap, tol in ((0.0, 100, 1e-6), (0.01, 25, 1e-6), (0.01, 100, None)):
        yg, qg, fg, ig, sg = h.solve_base_batch(QC, goals, ng, effort_weight=w, max_iter=cap)
        yo, qo, fo, io, so = o.solve_base_batch(QC, goals, ng, effort_weight=w, max_iter=cap)
        np.testing.assert_array_equal(ig, io)
        np.testing.assert_array_equal(sg, so)
        np.testing.assert_allclose(fg, fo, rtol=1e-6, atol=1e-14)
        if tol is not None:
            np.testing.assert_allclose(yg, yo, atol=tol)
            np.testing.assert_allclose(qg, qo, atol=tol)
        else:
            np.testing.assert_allclose(yg, yo, atol=5e-3)
        for b in range(B):  # padding rows carry the current configuration
            np.testing.assert_array_equal(qg[b, ng[b]:], np.broadcast_to(QC[b], (n - ng[b], desc.ndof)))
    # argument checks fail loudly
    with pytest.raises(capi.GTOError):
        h.solve_base_batch(QC, goals, np.zeros(B, dtype=np.int32))
    with pytest.raises(capi.GTOError):
        h.solve_base_batch(QC[:1], np.tile(np.eye(4), (1, 33, 1, 1)))
    h.close()


def test_depth_cost_field_matches_reference_golden_and_oracle(capi, oracle_mod):
    """gto_depth_sdf_cost through the DepthPointCloud surface: bit-identical to the reference-generated
    fixture, and to the CPU restatement on a larger random scene (ragged sizes, masked and invalid
    pixels, queries outside the viewport and behind the camera)."""
    import grasptrajopt_amd as g_
    g = golden("depth_cost.npz")
    for tag, tm in (("all", None), ("obs", g["mask"])):
        dpc = g_.DepthPointCloud(g["depth"], g["K"], g["cam"], target_mask=tm, threshold=1.5)
        q = g[f"{tag}_query"]
        np.testing.assert_array_equal(dpc.points, g[f"{tag}_points"])
        np.testing.assert_array_equal(dpc.get_sdf(q), g[f"{tag}_sdf"])
        np.testing.assert_array_equal(~dpc.is_outside(q), g[f"{tag}_inside"])
        np.testing.assert_array_equal(dpc.get_sdf_cost(q, epsilon=0.02, w_inside=1), g[f"{tag}_cost"])
    rng = np.random.default_rng(3)
    H, W = 117, 203
    K = np.array([[180.0, 0.3, 101.2], [0, 178.5, 58.7], [0, 0, 1.0]])
    depth = (0.6 + 0.5 * rng.random((H, W))).astype(np.float32)
    depth[rng.random((H, W)) < 0.1] = 0.0  # invalid
    depth[5:9, 7:40] = 2.0                 # beyond the threshold
    mask = (rng.random((H, W)) < 0.05).astype(np.uint8)
    a = 0.3
    cam = np.eye(4)
    cam[:3, :3] = np.array([[0, -np.sin(a), np.cos(a)], [-1.0, 0, 0], [0, -np.cos(a), -np.sin(a)]])
    cam[:3, 3] = [-0.3, 0.1, 0.8]
    q = rng.uniform([-1.0, -1.0, -0.5], [1.5, 1.0, 1.5], size=(1237, 3))
    dpc = g_.DepthPointCloud(depth, K, cam, target_mask=mask, threshold=1.5)
    pts, sdf, inside, cost = oracle_mod.depth_sdf_cost(depth, K, cam, mask, 1.5, q, epsilon=0.03, w_inside=2.0)
    np.testing.assert_array_equal(dpc.points, pts)
    np.testing.assert_array_equal(dpc.get_sdf(q), sdf)
    np.testing.assert_array_equal(~dpc.is_outside(q), inside)
    np.testing.assert_array_equal(dpc.get_sdf_cost(q, epsilon=0.03, w_inside=2.0), cost)
    assert inside.any() and (~inside).any() and (cost > 0).any()
    np.testing.assert_array_equal(dpc.get_sdf_in_batches(q, batch_size=500), sdf)
    assert dpc.get_sdf(np.zeros((0, 3))).shape == (0,)


@pytest.mark.parametrize("scene", ["noise", "surfaces"])
def test_depth_cost_field_tree_search_equals_exhaustive_search(capi, monkeypatch, scene):
    """The nearest-neighbour search of gto_depth_sdf_cost walks a bounding-box hierarchy over tiles of the depth image
    (k_depth_sdf_bvh); GTO_DEPTH_BRUTE=1 keeps the exhaustive search (k_depth_sdf, the reference construction).  Signed
    distances, inside flags and costs must be the same bits: every voxel centre of a grid around the cloud, a ragged
    image (sizes that are no multiple of the tile), invalid pixels and a masked target."""
    import grasptrajopt_amd as g_
    rng = np.random.default_rng(11)
    H, W = 123, 181
    K = np.array([[160.0, 0.0, 90.3], [0, 161.0, 61.1], [0, 0, 1.0]])
    if scene == "noise":
        depth = (0.6 + 0.5 * rng.random((H, W))).astype(np.float32)
    else:  # a floor plane seen at an angle with two boxes on it: what a depth camera sees
        v, u = np.mgrid[0:H, 0:W]
        depth = (0.9 + 0.002 * (v - H / 2) + 0.0005 * (u - W / 2)).astype(np.float32)
        depth[40:70, 50:90] -= 0.15
        depth[75:100, 110:150] -= 0.07
    depth[rng.random((H, W)) < 0.07] = 0.0
    mask = np.zeros((H, W), np.uint8)
    mask[45:60, 55:80] = 1
    a = 0.4
    cam = np.eye(4)
    cam[:3, :3] = np.array([[0, -np.sin(a), np.cos(a)], [-1.0, 0, 0], [0, -np.cos(a), -np.sin(a)]])
    cam[:3, 3] = [-0.3, 0.05, 0.8]
    ax = np.linspace(-0.6, 1.6, 40)
    q = np.stack(np.meshgrid(ax, ax - 0.5, ax - 0.4, indexing="ij"), -1).reshape(-1, 3)
    out = []
    for brute in ("0", "1"):
        monkeypatch.setenv("GTO_DEPTH_BRUTE", brute)
        dpc = g_.DepthPointCloud(depth, K, cam, target_mask=mask, threshold=1.4)
        out.append((dpc.get_sdf(q), ~dpc.is_outside(q), dpc.get_sdf_cost(q, epsilon=0.05, w_inside=1.5)))
    for x, y in zip(*out):
        np.testing.assert_array_equal(x, y)
    assert out[0][1].any() and (out[0][2] > 0).any()


# ------------------------------------------------------------------------------------------ assembled objectives
# tests/golden/objective.npz holds what the reference's OWN setup_optimization code computes (executed with numeric
# stand-ins for CasADi, tests/golden/make_objective_golden.py): these tests tie the HIP path to the reference
# directly, not through the oracle.
def _fixture_handle(capi, robot, g, kind=None):
    from grasptrajopt_amd.robot_desc import load_builtin
    cfg = cfg_of(robot)
    d = load_builtin(robot)
    np.testing.assert_array_equal(g[f"{robot}_points_checksum"], [d.points.sum(), np.abs(d.points).sum()])
    h = capi.SolverHandle(d, cfg["link_ee"], cfg["link_gripper"], device=0)
    if kind:
        h.set_scene(0, g[f"{robot}_field_{kind}_all"], g[f"{robot}_field_{kind}_obs"], g[f"{robot}_grid_shape"],
                    g[f"{robot}_grid_origin"], float(g[f"{robot}_grid_res"]))
    return h


@pytest.mark.parametrize("mode", [0])
@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_objective_terms_vs_reference_fixture(capi, robot, mode):
    """gto_eval_objective against the reference's cost expressions (gto/gto_planner.py:84-135): f_goal of the set and of
    every goal alone, arg-min goal (bit-exact), f_obs on a dense random field (every point's voxel matters) and on a
    sparse one, f_vel.  1e-10 relative."""
    g = golden("objective.npz")
    Q = g[f"{robot}_Q"]
    for tag in [str(c) for c in g["cases"] if str(c).startswith(robot)]:
        h = _fixture_handle(capi, robot, g, tag.split("_")[-1])
        h.set_mode(mode)
        RT = g[tag + "_RT"]
        n = RT.shape[1]
        so = g[f"{robot}_standoff"] if "_so1_" in tag else None
        each = g[tag + "_f_goal_each"]
        fg, fo, fv, am = h.eval_objective(0, RT.reshape(len(Q), n, 16), n, so, g[tag + "_base"], Q)
        np.testing.assert_allclose(fg, each.min(axis=1), rtol=1e-10, err_msg=tag)
        np.testing.assert_array_equal(am, each.argmin(axis=1), err_msg=tag)
        np.testing.assert_allclose(fo, g[tag + "_f_obs"], rtol=1e-10, err_msg=tag)
        np.testing.assert_allclose(fv, g[tag + "_f_vel"], rtol=1e-10, err_msg=tag)
        for k in range(n):
            fk, _, _, _ = h.eval_objective(0, RT[:, k].reshape(len(Q), 1, 16), 1, so, g[tag + "_base"], Q)
            np.testing.assert_allclose(fk, each[:, k], rtol=1e-10, err_msg=f"{tag} goal {k}")
        h.close()


@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_ik_objective_vs_reference_fixture(capi, robot):
    """gto_solve_ik_batch capped at 0 iterations returns the objective at its seed: gto/ik_solver.py:46-70."""
    g = golden("objective.npz")
    h = _fixture_handle(capi, robot, g, "dense")
    q, RT, base = g[f"{robot}_ik_q"], g[f"{robot}_ik_RT"].reshape(-1, 16), g[f"{robot}_ik_base"]
    qo, cost, it, st = h.solve_ik_batch(0, q, RT, base, max_iter=0)
    np.testing.assert_array_equal(qo, q)
    np.testing.assert_allclose(cost, g[f"{robot}_ik_cost_pos"] + g[f"{robot}_ik_cost_obstacle"], rtol=1e-10)
    _, cost0, _, _ = h.solve_ik_batch(None, q, RT, None, max_iter=0)
    np.testing.assert_allclose(cost0, g[f"{robot}_ik_cost_pos"], rtol=1e-10)
    h.close()


@pytest.mark.parametrize("robot", ["panda", "fetch"])
@pytest.mark.parametrize("n", [1, 3])
def test_base_objective_vs_reference_fixture(capi, robot, n):
    """gto_eval_base_objective (the base-placement kernel started at (y, q) and capped at 0 iterations) against
    gto/base_planner.py:44-87, both the goal_size == 1 and the goal-set branch."""
    g = golden("objective.npz")
    h = _fixture_handle(capi, robot, g)
    y, Qb, RT = g[f"{robot}_base_n{n}_y"], g[f"{robot}_base_n{n}_Q"], g[f"{robot}_base_n{n}_RT"]
    q = np.transpose(Qb, (0, 2, 1))
    eff, pos = g[f"{robot}_base_n{n}_cost_effort"], g[f"{robot}_base_n{n}_cost_pos"]
    np.testing.assert_allclose(h.eval_base_objective(y, q, RT, effort_weight=0.0), pos, rtol=1e-10)
    np.testing.assert_allclose(h.eval_base_objective(y, q, RT, effort_weight=0.01), pos + eff, rtol=1e-10)
    h.close()


# ------------------------------------------------------------------------------------------ BASELINE configs at size
def _invariants(desc, qc, Q0, Q, dQ):
    oi = desc.opt_index
    assert (Q[:, oi] >= desc.lower[oi][None, :, None]).all() and (Q[:, oi] <= desc.upper[oi][None, :, None]).all()
    assert np.abs(Q[:, :, 1] - Q[:, :, 0]).max() == 0.0
    np.testing.assert_array_equal(Q[:, oi, 0], qc[:, oi])
    np.testing.assert_array_equal(Q[:, desc.param_index], Q0[:, desc.param_index])
    T = Q.shape[2]
    np.testing.assert_allclose(Q[:, :, :-1] + (10.0 / (T - 1)) * dQ, Q[:, :, 1:], atol=1e-15)


def test_baseline_config3_one_gpus_share_256_scenes_resident(capi, oracle_mod):
    """BASELINE configs[3] at the size ONE of its eight GPUs carries: 256 different 128^3 scenes resident at once (fields,
    voxel records and distance fields of every scene: 148 MB each, 37.9 GB in all -- one field per scene as in
    gto/gto_models.py:155-171), 8 grasps per scene = 2048 instances of the Panda-5k workload in ONE solve call through
    solve_local_shard, every instance of a call reading another field.  Resident memory against hipMemGetInfo, invariants
    and the objective recomputed on all 2048, the oracle on a sample of six instances from six scenes."""
    import torch
    from grasptrajopt_amd import synthetic as syn
    from grasptrajopt_amd.parallel import shard_by_scene, solve_local_shard
    from grasptrajopt_amd.robot_desc import load_builtin
    cfg = cfg_of("panda")
    d = load_builtin("panda_5k")
    opts = oracle_mod.reference_opts(max_iter=100)
    h = capi.SolverHandle(d, cfg["link_ee"], cfg["link_gripper"], opts, device=0, n_gripper_points=100)
    NS, G, T, N = 256, 8, 50, 128
    free0 = torch.cuda.mem_get_info(0)[0]
    moving = d.link_is_moving()[d.point_link]
    RT, qg, keep = [], [], {}
    sel = [5, 411, 1000, 1337, 1799, 2047]
    for s in range(NS):
        sc = syn.make_scene(100 + s, n=N, res=2.24 / N)
        h.set_scene(s, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
        if s in [i // G for i in sel]:
            keep[s] = sc

        def cc(q, s=s):
            _, _, val, _ = h.eval_points(s, q, [0.0, 0.0, 0.0], use_obs=True)
            return (val * moving[None, :]).sum(axis=1)
        r, q = syn.make_goals(d, h.eval_fk, cfg["link_ee"], G, seed=7000 + s, collision_cost=cc)
        RT.append(r)
        qg.append(q)
    resident = free0 - torch.cuda.mem_get_info(0)[0]
    per_scene = 2 * (N ** 3) * (4 + 32 + 1)  # c_all / c_obs: float32 field + 32-B voxel record + distance byte per voxel
    assert NS * per_scene <= resident <= 1.08 * NS * per_scene + (1 << 30), (resident / 1e9, NS * per_scene / 1e9)
    assert 37.0 <= NS * per_scene / 1e9 <= 40.5
    RT, qg = np.concatenate(RT), np.concatenate(qg)
    nI = NS * G
    sid = np.repeat(np.arange(NS, dtype=np.int32), G)
    qc = np.tile(np.array(cfg["default_pose"], dtype=np.float64), (nI, 1))
    Q0 = np.stack([syn.make_seed(qc[i], qg[i], T, d.param_index) for i in range(nI)])
    S = np.tile(syn.standoff_pose(-0.1, cfg["axis_standoff"]).reshape(1, 16), (nI, 1))
    base = np.zeros((nI, 3))
    owner = shard_by_scene(sid, 1)
    idx, Q, dQ, f, it, st = solve_local_shard(h.solve_batch, np.arange(nI), sid, qc, RT.reshape(nI, 1, 16), np.ones(nI, np.int32), S, base, Q0,
                                              B=nI, rank=0, world=1, assignment=owner)
    np.testing.assert_array_equal(idx, np.arange(nI))
    _invariants(d, qc, Q0, Q, dQ)
    assert set(np.unique(st).tolist()) <= {0, 1} and (st == 0).mean() > 0.98
    # the objective the solver reports is the objective of the trajectory it returns, on every instance; never above the seed's
    for lo in range(0, nI, 512):
        sl = slice(lo, lo + 512)
        fg, fo_, fv, _ = h.eval_objective(sid[sl], RT[sl].reshape(-1, 1, 16), 1, S[sl], base[sl], Q[sl])
        np.testing.assert_allclose(fg + fo_ + fv, f[sl], rtol=1e-9, atol=1e-12)
        seed = Q0[sl].copy()
        oi = d.opt_index
        seed[:, oi] = np.clip(seed[:, oi], d.lower[oi][None, :, None], d.upper[oi][None, :, None])
        seed[:, :, :2] = qc[sl][:, :, None]
        sg, so_, sv, _ = h.eval_objective(sid[sl], RT[sl].reshape(-1, 1, 16), 1, S[sl], base[sl], seed)
        assert (f[sl] <= (sg + so_ + sv) * (1 + 1e-12)).all()
    o = oracle_mod.Oracle(d, cfg["link_ee"], cfg["link_gripper"], opts, n_gripper_points=100)
    for s, sc in keep.items():
        o.set_scene(s, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    Qo, _, fo, ito, sto = o.solve_batch(sid[sel], qc[sel], RT[sel].reshape(-1, 1, 16), 1, S[sel], base[sel], Q0[sel])
    np.testing.assert_array_equal(it[sel], ito)
    np.testing.assert_array_equal(st[sel], sto)
    np.testing.assert_allclose(Q[sel], Qo, rtol=0, atol=1e-6)
    np.testing.assert_allclose(f[sel], fo, rtol=1e-7)
    h.close()


@pytest.mark.parametrize("mode", [0])
def test_fetch_shelf_256_instances(capi, oracle_mod, mode):
    """BASELINE configs[2] at size: Fetch arm, shelf scene, 256 (scene, grasp) instances, T = 50, 128^3 field at the bench's
    resolution, the reference's iteration cap of 100.  Shelf scenes are planned from interpolate=False seeds
    (gto/gto_planner.py:216-219, examples/pybullet_gto_planning.py:98-109): the arm holds qc until the standoff waypoint
    and jumps to the IK solution.  Oracle on a sample of seven (every usable core), size-independent properties on all 256."""
    from grasptrajopt_amd import synthetic as syn
    from grasptrajopt_amd.robot_desc import load_builtin
    cfg = cfg_of("fetch")
    d = load_builtin("fetch")
    opts = oracle_mod.reference_opts(max_iter=100)
    h = capi.SolverHandle(d, cfg["link_ee"], cfg["link_gripper"], opts, device=0)
    h.set_mode(mode)
    sc = syn.make_scene(11, n=128, res=2.24 / 128, origin=(-0.3, -1.12, 0.0), table_z=0.75, shelf=True)
    assert sc.objects[-1][0] == "shelf"
    h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    B, T = 256, 50
    moving = d.link_is_moving()[d.point_link]

    def cc(q):
        _, _, val, _ = h.eval_points(0, q, [0.0, 0.0, 0.0], use_obs=True)
        return (val * moving[None, :]).sum(axis=1)
    RT, qg = syn.make_goals(d, h.eval_fk, cfg["link_ee"], B, seed=5, xlim=(0.45, 0.85), ylim=(-0.45, 0.45), zlim=(0.82, 1.08), collision_cost=cc)
    qc = np.tile(np.array(cfg["default_pose"], dtype=np.float64), (B, 1))
    qg[:, d.param_index] = qc[:, d.param_index]
    Q0 = np.repeat(qc[:, :, None], T, axis=2)
    Q0[:, :, T - 10:] = qg[:, :, None]  # interpolate=False
    S = syn.standoff_pose(-0.1, cfg["axis_standoff"])
    args = (0, qc, RT.reshape(B, 1, 16), 1, S, [0.0, 0.0, 0.0], Q0)
    Q, dQ, f, it, st = h.solve_batch(*args)
    _invariants(d, qc, Q0, Q, dQ)
    fg, fo, fv, _ = h.eval_objective(0, RT.reshape(B, 1, 16), 1, S, [0.0, 0.0, 0.0], Q)
    np.testing.assert_allclose(fg + fo + fv, f, rtol=1e-10)  # reported cost == objective at the returned trajectory
    seed = Q0.copy()
    oi = d.opt_index
    seed[:, oi] = np.clip(seed[:, oi], d.lower[oi][None, :, None], d.upper[oi][None, :, None])
    sg, so, sv, _ = h.eval_objective(0, RT.reshape(B, 1, 16), 1, S, [0.0, 0.0, 0.0], seed)
    assert (f <= (sg + so + sv) * (1 + 1e-12)).all()
    assert len(set(it.tolist())) > 3  # a real mix of easy and hard instances
    o = oracle_mod.Oracle(d, cfg["link_ee"], cfg["link_gripper"], opts)
    o.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    sel = [0, 31, 64, 100, 129, 200, 255]
    Qo, _, fo_, ito, sto = o.solve_batch(0, qc[sel], RT[sel].reshape(-1, 1, 16), 1, S, [0.0, 0.0, 0.0], Q0[sel], n_threads=o.usable_cores())
    np.testing.assert_array_equal(it[sel], ito)
    np.testing.assert_array_equal(st[sel], sto)
    np.testing.assert_allclose(Q[sel], Qo, rtol=0, atol=1e-6)
    h.close()


def test_scene_sharded_many_scenes(capi, oracle_mod):
    """BASELINE configs[3] on one rank: 64 different scenes x 8 grasps through parallel.solve_sharded with
    SolverHandle.solve_batch (instances grouped by scene, every scene resident once); every instance of the call reads
    another field.  Same results as one plain call over everything; oracle on a sample."""
    from grasptrajopt_amd import synthetic as syn
    from grasptrajopt_amd.parallel import shard_by_scene, solve_sharded
    from grasptrajopt_amd.robot_desc import load_builtin
    cfg = cfg_of("panda")
    d = load_builtin("panda")
    opts = oracle_mod.reference_opts(max_iter=25)
    h = capi.SolverHandle(d, cfg["link_ee"], cfg["link_gripper"], opts, device=0)
    o = oracle_mod.Oracle(d, cfg["link_ee"], cfg["link_gripper"], opts)
    NS, G, T = 64, 8, 50
    scenes = []
    RT, qg = [], []
    for s in range(NS):
        sc = syn.make_scene(200 + s, n=48, res=0.0467)
        scenes.append(sc)
        h.set_scene(s, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
        r, q = syn.make_goals(d, h.eval_fk, cfg["link_ee"], G, seed=300 + s)
        RT.append(r)
        qg.append(q)
    RT, qg = np.concatenate(RT), np.concatenate(qg)
    nI = NS * G
    sid = np.repeat(np.arange(NS, dtype=np.int32), G)
    qc = np.tile(np.array(cfg["default_pose"], dtype=np.float64), (nI, 1))
    Q0 = np.stack([syn.make_seed(qc[i], qg[i], T, d.param_index) for i in range(nI)])
    S = syn.standoff_pose(-0.1, cfg["axis_standoff"])
    owner = shard_by_scene(sid, 1)
    idx, Q, dQ, f, it, st = solve_sharded(h.solve_batch, sid, qc, RT.reshape(nI, 1, 16), 1, S, [0.0, 0.0, 0.0], Q0, rank=0, world=1,
                                          assignment=owner)
    np.testing.assert_array_equal(idx, np.arange(nI))
    _invariants(d, qc, Q0, Q, dQ)
    direct = h.solve_batch(sid, qc, RT.reshape(nI, 1, 16), 1, S, [0.0, 0.0, 0.0], Q0)
    np.testing.assert_array_equal(Q, direct[0])
    np.testing.assert_array_equal(it, direct[3])
    sel = [3, 77, 130, 255, 300, 511]
    for s in sorted(set(sid[sel].tolist())):
        sc = scenes[s]
        o.set_scene(s, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    Qo, _, fo, ito, _ = o.solve_batch(sid[sel], qc[sel], RT[sel].reshape(-1, 1, 16), 1, S, [0.0, 0.0, 0.0], Q0[sel])
    np.testing.assert_array_equal(it[sel], ito)
    np.testing.assert_allclose(Q[sel], Qo, rtol=0, atol=1e-6)
    np.testing.assert_allclose(f[sel], fo, rtol=1e-7)
    h.close()


@pytest.mark.parametrize("robot,n_goals,T,off,grad", [("panda", 1, 50, -10, 0), ("fetch", 3, 50, -10, 0), ("panda_5k", 2, 30, -6, 0),
                                                      ("panda", 4, 50, -10, 1), ("fetch", 1, 64, -12, 0)])
def test_parity_sweep_reduced(capi, oracle_mod, robot, n_goals, T, off, grad):
    """tools/parity_sweep.py at a size that fits the test run: 3 scenes x 16 instances per configuration (robots, goal-set
    sizes, horizons, both gradient modes): identical iteration counts and status, trajectories within 1e-6 rad."""
    nthr = oracle_mod.Oracle.usable_cores()
    for seed in (21, 22, 23):
        prob = Problem(robot, B=16, scene_seed=seed, n_goals=n_goals, T=T)
        opts = oracle_mod.reference_opts(T=T, standoff_offset=off, grad_mode=grad, max_iter=60)
        h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
        o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
        prob.finish(h.eval_fk)
        h.set_scene(*prob.scene_args())
        o.set_scene(*prob.scene_args())
        Qg, _, fg, itg, stg = h.solve_batch(*prob.solve_args())
        Qo, _, fo, ito, sto = o.solve_batch(*prob.solve_args(), n_threads=nthr)
        np.testing.assert_array_equal(itg, ito, err_msg=f"seed {seed}")
        np.testing.assert_array_equal(stg, sto, err_msg=f"seed {seed}")
        np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-6, err_msg=f"seed {seed}")
        h.close()


@pytest.mark.parametrize("mode", [0])
def test_device_entry_point_on_caller_stream(capi, oracle_mod, mode):
    """gto_solve_batch_device called directly: every array resident in HBM (torch tensors), work enqueued on the caller's
    stream, results valid after that stream is synchronised; equal to the host-pointer call."""
    import torch
    prob = Problem("panda", B=10, scene_seed=4, n_goals=2)
    h, o = make_pair(capi, oracle_mod, prob, mode=mode, max_iter=20)
    ref = h.solve_batch(*prob.solve_args())
    dev = torch.device("cuda", 0)
    B, d, T = prob.B, prob.desc, prob.T
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    inp = [torch.zeros(B, dtype=torch.int32, device=dev), t(prob.qc, torch.float64), t(prob.goals, torch.float64),
           torch.full((B,), 2, dtype=torch.int32, device=dev), t(np.tile(prob.S.reshape(1, 16), (B, 1)), torch.float64),
           t(prob.base, torch.float64), t(prob.Q0, torch.float64)]
    out = [torch.empty((B, d.ndof, T), dtype=torch.float64, device=dev), torch.empty((B, d.ndof, T - 1), dtype=torch.float64, device=dev),
           torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)]
    stream = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    h.solve_batch_device(B, 2, *[x.data_ptr() for x in inp + out], stream.cuda_stream)
    stream.synchronize()
    for a, b in zip(ref, out):
        np.testing.assert_array_equal(a, b.cpu().numpy())
    # outputs are optional
    h.solve_batch_device(B, 2, *[x.data_ptr() for x in inp], out[0].data_ptr(), None, None, None, None, stream.cuda_stream)
    stream.synchronize()
    np.testing.assert_array_equal(ref[0], out[0].cpu().numpy())
    h.close()


# ------------------------------------------------------------------------------------------ more than eight optimised joints
@pytest.mark.parametrize("T,off,n_goals,grad", [(50, -10, 1, 0), (50, -10, 3, 0), (80, -16, 2, 0), (50, -10, 1, 1)])
def test_mobile_fetch_ten_joints_matches_oracle(capi, oracle_mod, T, off, n_goals, grad):
    """Fetch arm on a planar base (robot_desc.with_planar_base): 3 base + 7 arm = 10 optimised joints, the 16-wide blocks
    (k_obstacle_gram<16>, k_lm_step_wide).  Objective terms, normal equations and the solve against the oracle."""
    prob = Problem("fetch_mobile", B=6, scene_seed=5, n_goals=n_goals, T=T, shelf=True, table_z=0.75)
    assert prob.desc.n_opt == 10
    opts = oracle_mod.reference_opts(T=T, standoff_offset=off, grad_mode=grad, max_iter=40)
    h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    prob.finish(h.eval_fk)
    h.set_scene(*prob.scene_args())
    o.set_scene(*prob.scene_args())
    np.testing.assert_allclose(h.eval_fk(prob.qgoal[:, 0]), o.eval_fk(prob.qgoal[:, 0]), rtol=0, atol=1e-12)
    eg = h.eval_objective(0, prob.goals, n_goals, prob.S, prob.base, prob.Q0)
    eo = o.eval_objective(0, prob.goals, n_goals, prob.S, prob.base, prob.Q0)
    for a, b in zip(eg[:3], eo[:3]):
        np.testing.assert_allclose(a, b, rtol=1e-10)
    np.testing.assert_array_equal(eg[3], eo[3])
    Jg, jg, sg = h.eval_obstacle_normal_eq(0, prob.base, prob.Q0)
    Jo, jo, so = o.eval_obstacle_normal_eq(0, prob.base, prob.Q0)
    sc = max(np.abs(Jo).max(), 1e-30)
    np.testing.assert_allclose(Jg, Jo, rtol=0, atol=1e-10 * sc)
    np.testing.assert_allclose(jg, jo, rtol=0, atol=1e-10 * max(np.abs(jo).max(), 1e-30))
    np.testing.assert_allclose(sg, so, rtol=1e-10, atol=1e-300)
    Qg, dQg, fg, itg, stg = h.solve_batch(*prob.solve_args())
    Qo, dQo, fo, ito, sto = o.solve_batch(*prob.solve_args())
    np.testing.assert_array_equal(itg, ito)
    np.testing.assert_array_equal(stg, sto)
    np.testing.assert_allclose(Qg, Qo, rtol=0, atol=1e-6)
    np.testing.assert_allclose(fg, fo, rtol=1e-7)
    _invariants(prob.desc, prob.qc, prob.Q0, Qg, dQg)
    # the entry points that only exist for up to eight optimised joints say so
    with pytest.raises(capi.GTOError, match="eight"):
        h.solve_ik_batch(None, prob.qc, prob.goals[:, 0], None, 5)
    with pytest.raises(capi.GTOError, match="eight"):
        h.solve_base_batch(prob.qc, prob.goals[:, :1].reshape(-1, 1, 4, 4))
    with pytest.raises(capi.GTOError, match="eight"):  # ... the evaluation twin of the base solve as well (8-joint LDS layout)
        h.eval_base_objective(np.zeros((6, 3)), prob.qc.reshape(6, 1, -1), prob.goals[:, :1].reshape(-1, 1, 4, 4), effort_weight=0.0)
    # seed scoring and point look-ups run the kinematics of all ten joints (screw table sized for the 16-wide blocks)
    pc_g, dist_g = h.plan_cost(0, prob.Q0, prob.base[0])
    pc_o, dist_o = o.plan_cost(0, prob.Q0, prob.base[0])
    np.testing.assert_allclose(pc_g, pc_o, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(dist_g, dist_o, rtol=1e-14)
    xg, og, vg, _ = h.eval_points(0, prob.qgoal[:, 0], prob.base[0], use_obs=True)
    xo, oo, vo, _ = o.eval_points(0, prob.qgoal[:, 0], prob.base[0], use_obs=True)
    np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(og, oo)
    np.testing.assert_array_equal(vg, vo)
    with pytest.raises(capi.GTOError, match="removed"):  # the single-launch mode of rounds 1-3 is gone; its number stays reserved
        h.set_mode(1)
    h.close()


def test_mobile_fetch_baseline_config4_size(capi, oracle_mod):
    """BASELINE configs[4]: mobile Fetch, 10 optimised joints, T = 80 waypoints, 256^3 cost field (1.2 GB resident with its
    voxel records and distance fields), shelf scene.  Oracle on a sample, invariants and objective consistency on all."""
    from grasptrajopt_amd import synthetic as syn
    T, B = 80, 64
    prob = Problem("fetch_mobile", B=B, scene_seed=8, n=256, res=0.0175, T=T, shelf=True, table_z=0.75, scene_origin=(-1.6, -2.24, -0.2))
    assert prob.scene.shape == (256, 256, 256)
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-16, max_iter=100)  # the bench's batch of 64 at the reference's cap
    h = capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    prob.finish(h.eval_fk)
    h.set_scene(*prob.scene_args())
    Q, dQ, f, it, st = h.solve_batch(*prob.solve_args())
    _invariants(prob.desc, prob.qc, prob.Q0, Q, dQ)
    fg, fo, fv, _ = h.eval_objective(0, prob.goals, 1, prob.S, prob.base, Q)
    np.testing.assert_allclose(fg + fo + fv, f, rtol=1e-10)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    o.set_scene(*prob.scene_args())
    sel = [0, 7, 15, 40, 63]
    Qo, _, fo_, ito, sto = o.solve_batch(0, prob.qc[sel], prob.goals[sel], 1, prob.S, prob.base[sel], prob.Q0[sel], n_threads=o.usable_cores())
    np.testing.assert_array_equal(it[sel], ito)
    np.testing.assert_array_equal(st[sel], sto)
    np.testing.assert_allclose(Q[sel], Qo, rtol=0, atol=1e-6)
    h.close()


# ------------------------------------------------------------------------------------------ non-finite inputs
@pytest.mark.parametrize("robot,T", [("panda", 50), ("fetch_mobile", 20)])
def test_non_finite_inputs_end_in_status_numerical(capi, oracle_mod, robot, T):
    """The reference returns the iterate whatever the solver says (optas/solver.py:135, error_on_fail=False); here a solve
    whose seed has no finite objective ends with GTO_STATUS_NUMERICAL and 0 iterations for exactly that instance.  NaN / Inf
    in a goal pose (alone: fails; next to a finite goal of the set: never the arg-min, the solve is the solve without it), in
    the seed (NaN: fails; -Inf: the clip to the joint limits makes it a number) and in a voxel the seed touches: HIP path
    and oracle agree on status and iteration count of every instance, and the clean neighbours in the same batch get the
    bits they get in a batch without the poisoned ones (narrow step kernels on the Panda, the wide one on the ten-joint
    mobile Fetch)."""
    B = 14
    prob = Problem(robot, B=B, scene_seed=3, T=T, n_goals=2)
    kw = dict(max_iter=25, T=T, standoff_offset=-max(2, T // 5))
    h, o = make_pair(capi, oracle_mod, prob, **kw)
    d = prob.desc
    ng_clean = np.full(B, 2, dtype=np.int32)
    ng_clean[4] = 1                              # (what instance 4 of the poisoned batch must come out as)
    clean = (0, prob.qc, prob.goals, ng_clean, prob.S, prob.base, prob.Q0)
    Qc, dQc, fc, itc, stc = h.solve_batch(*clean)
    assert set(np.unique(stc).tolist()) <= {0, 1}
    goals, Q0 = prob.goals.copy(), prob.Q0.copy()
    sid = np.zeros(B, dtype=np.int32)
    ng = np.full(B, 2, dtype=np.int32)
    goals[2, 0, 3] = np.nan                    # translation of the only goal of the set
    ng[2] = 1
    goals[4, 1, 5] = np.inf                    # rotation entry of the SECOND goal: the first one is finite
    goals[12, 0, 7] = np.nan                   # both goals of the set
    goals[12, 1, 0] = -np.inf
    Q0[7, d.opt_index[1], T // 2] = np.nan     # an optimised joint of the seed
    Q0[9, d.opt_index[0], 3] = -np.inf         # clipped to the lower limit: a number again
    # a poisoned voxel under the seed of instance 11: scene 1 = scene 0 with NaN in a record the seed's surface points touch
    seed_c = np.clip(Q0[11][d.opt_index], d.lower[d.opt_index][:, None], d.upper[d.opt_index][:, None])
    q_mid = Q0[11][:, T // 2].copy()
    q_mid[d.opt_index] = seed_c[:, T // 2]
    _, off, val, _ = h.eval_points(0, q_mid[None, :], prob.base[11], use_obs=False)
    moving = np.nonzero(d.link_is_moving()[d.point_link])[0]
    pick = int(moving[len(moving) // 2])
    c_all2 = np.array(prob.scene.c_all, dtype=np.float32).copy().reshape(-1)
    c_obs2 = np.array(prob.scene.c_obs, dtype=np.float32).copy().reshape(-1)
    c_all2[int(off[0, pick])] = c_obs2[int(off[0, pick])] = np.nan
    for s in (h, o):
        s.set_scene(1, c_all2, c_obs2, prob.scene.shape, prob.scene.origin, prob.scene.res)
    sid[11] = 1
    args = (sid, prob.qc, goals, ng, prob.S, prob.base, Q0)
    Qg, dQg, fg, itg, stg = h.solve_batch(*args)
    Qo, dQo, fo, ito, sto = o.solve_batch(*args)
    np.testing.assert_array_equal(stg, sto)
    np.testing.assert_array_equal(itg, ito)
    expect_fail = [2, 7, 11, 12]
    assert (stg[expect_fail] == capi.GTO_STATUS_NUMERICAL).all() and (itg[expect_fail] == 0).all(), (stg.tolist(), itg.tolist())
    ok = np.setdiff1d(np.arange(B), expect_fail)
    assert (stg[ok] != capi.GTO_STATUS_NUMERICAL).all()
    untouched = np.setdiff1d(ok, [9])           # (instance 4 included: a non-finite goal next to a finite one changes nothing)
    for a, b in ((Qg, Qc), (dQg, dQc), (fg, fc), (itg, itc), (stg, stc)):  # neighbours: the bits of the clean batch
        np.testing.assert_array_equal(a[untouched], b[untouched])
    np.testing.assert_allclose(Qg[ok], Qo[ok], rtol=0, atol=1e-6)
    assert np.isfinite(Qg[ok]).all() and np.isfinite(fg[ok]).all()
    # the handle is usable afterwards and gives the clean batch its bits again
    Q2, _, f2, it2, st2 = h.solve_batch(*clean)
    np.testing.assert_array_equal(Q2, Qc)
    np.testing.assert_array_equal(it2, itc)
    h.close()
