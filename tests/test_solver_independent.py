"""Independent checks of the solver iteration on the CPU oracle (see tests/independent.py)."""
import numpy as np
import pytest

from helpers import Problem
from independent import SmoothProblem, check_against_lbfgsb


@pytest.fixture(scope="module")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def _setup(oracle_mod, robot, T=12, B=4, n_goals=1):
    prob = Problem(robot, B=B, scene_seed=1, T=T, n_goals=n_goals)
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-3, grad_mode=1, max_iter=300, tol_rel_f=1e-14)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    prob.finish(o.eval_fk)
    sc = prob.scene
    zero = np.zeros_like(sc.c_all)  # the smooth problem: no obstacle term (its gradient is zero under CasADi AD anyway)
    o.set_scene(0, zero, zero, sc.shape, sc.origin, sc.res)
    return prob, opts, o


@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_gradient_blocks_are_the_derivative_of_the_objective(oracle_mod, robot):
    """2 J^T r of the goal-set, standoff and velocity terms against central differences of the objective value, at
    random interior trajectories: every Jacobian on the path (FK, gripper-cloud moments, screws) in one number."""
    prob, opts, o = _setup(oracle_mod, robot)
    rng = np.random.default_rng(0)
    for b in range(prob.B):
        sp = SmoothProblem(o, prob, b, opts, prob.Q0[b])
        x = sp.pack(prob.Q0[b]) + 0.05 * rng.standard_normal(sp.n * (sp.T - 2))
        x = np.clip(x, sp.lo_x + 1e-3, sp.hi_x - 1e-3)
        ga, gf = sp.grad(x), sp.grad_fd(x)
        assert np.abs(ga - gf).max() <= 1e-7 * max(1.0, np.abs(ga).max())


@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_oracle_lm_ends_where_lbfgsb_ends(oracle_mod, robot):
    prob, opts, o = _setup(oracle_mod, robot)
    Q, _, f, it, st = o.solve_batch(*prob.solve_args())
    check_against_lbfgsb(o, prob, opts, Q, f, st, min_same_basin=2)


@pytest.mark.parametrize("robot", ["panda", "fetch"])
def test_oracle_ik_ends_where_lbfgsb_ends(oracle_mod, robot):
    from independent import check_ik_against_lbfgsb
    prob = Problem(robot, B=8, scene_seed=5)
    opts = oracle_mod.reference_opts(tol_rel_f=1e-14)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    prob.finish(o.eval_fk)
    rng = np.random.default_rng(1)
    oi = prob.desc.opt_index
    q0 = prob.qc.copy()
    q0[4:, oi] = prob.qgoal[4:, 0][:, oi] + rng.uniform(-0.3, 0.3, size=(4, len(oi)))  # half far, half near seeds
    q, f, it, st = o.solve_ik_batch(None, q0, prob.goals[:, 0], None, max_iter=200)
    check_ik_against_lbfgsb(o, prob, q0, q, f, min_agree=6)


def test_oracle_base_placement_ends_where_lbfgsb_ends(oracle_mod):
    from independent import check_base_against_lbfgsb
    from grasptrajopt_amd import synthetic as syn
    prob = Problem("fetch", B=4, scene_seed=2)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], oracle_mod.reference_opts(tol_rel_f=1e-14))
    goals, _ = syn.make_base_goal_sets(prob.desc, o.eval_fk, prob.cfg["link_ee"], prob.qc[0], 4, 3, 0)
    y, q, f, it, st = o.solve_base_batch(prob.qc, goals, None, 1.0, max_iter=300)
    check_base_against_lbfgsb(o, prob.desc, prob.qc, np.asarray(goals).reshape(4, 3, 16), 1.0, y, q, f, min_agree=3)


@pytest.mark.parametrize("robot,grad_mode", [("panda", 0), ("fetch", 0)])
def test_oracle_obstacle_blocks_against_finite_difference_jacobians(oracle_mod, robot, grad_mode):
    from independent import check_obstacle_blocks_against_fd
    T = 16
    prob = Problem(robot, B=2, scene_seed=3, T=T)
    opts = oracle_mod.reference_opts(T=T, standoff_offset=-4, grad_mode=grad_mode)
    o = oracle_mod.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts)
    prob.finish(o.eval_fk)
    o.set_scene(*prob.scene_args())
    nz = sum(check_obstacle_blocks_against_fd(o, prob.desc, T, T - 4, prob.Q0[b], prob.base[b]) for b in range(prob.B))
    assert nz >= 4  # the trajectories do pass through the cost band
