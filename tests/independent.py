"""Checks of the solver iteration that use neither the oracle's iteration code nor its derivative code as the judge.

The reference's IPOPT run has no stored output and IPOPT is replaced on purpose, so the iterates
cannot be pinned.  What can be checked independently is where the iteration ENDS, on the problem IPOPT actually sees:
under CasADi AD the obstacle term has an identically zero gradient, i.e. the gradient IPOPT
follows is that of goal-set matching + standoff + velocity smoothness under the joint limits.  On that smooth
problem (GTO_GRAD_ZERO, empty cost field):
  * the analytic gradient assembled from the normal-equation blocks (2 J^T r) is compared with central finite
    differences of the objective VALUE (a derivative check of every Jacobian on the path);
  * SciPy's L-BFGS-B (a third-party bound-constrained optimiser), started from the same seed, is run on the same
    objective; where it ends in the same basin, its minimiser and minimum must be the solver's.
"""
import numpy as np


class SmoothProblem:
    """One instance of a tests.helpers.Problem as a function of the free variables x = Q[opt, 2:] (row-major)."""

    def __init__(self, oracle_obj, prob, b, opts, Q_template):
        self.o, self.prob, self.b = oracle_obj, prob, b
        d = prob.desc
        self.oi, self.T = d.opt_index, opts.T
        self.n = len(self.oi)
        self.lo, self.hi = d.lower[self.oi], d.upper[self.oi]
        self.w_obs = opts.w_obstacle
        self.alpha = opts.w_vel / ((opts.Tmax / (opts.T - 1)) ** 2)
        self.ts = opts.T + opts.standoff_offset
        self.Qfull = np.array(Q_template, dtype=np.float64)  # pinned waypoints 0, 1 and the parameter joints as the solver has them
        self.lo_x, self.hi_x = np.repeat(self.lo, self.T - 2), np.repeat(self.hi, self.T - 2)

    def unpack(self, x):
        Q = self.Qfull.copy()
        Q[self.oi, 2:] = np.asarray(x).reshape(self.n, self.T - 2)
        return Q

    def pack(self, Q):
        return np.asarray(Q)[self.oi, 2:].reshape(-1).copy()

    def f(self, X):
        """Objective of k points X [k, n (T-2)] in one oracle call."""
        X = np.atleast_2d(X)
        Qb = np.stack([self.unpack(x) for x in X])
        k, p, b = len(Qb), self.prob, self.b
        fg, fo, fv, _ = self.o.eval_objective(0, np.repeat(p.goals[b:b + 1], k, 0), p.n_goals, p.S, np.repeat(p.base[b:b + 1], k, 0), Qb)
        return fg + self.w_obs * fo + fv

    def grad(self, x):
        """2 J^T r from the normal-equation blocks + the velocity terms (what the LM step is built from)."""
        p, b, n, T = self.prob, self.b, self.n, self.T
        Q = self.unpack(x)
        _, Jtr, _, _, gg = self.o.eval_normal_eq(0, p.goals[b:b + 1], p.n_goals, p.S, p.base[b:b + 1], Q[None])
        q, g = Q[self.oi], np.zeros((n, T))
        for t in range(2, T):
            bt = self.w_obs * Jtr[0, t].copy()
            if t == T - 1:
                bt += gg[0, 0]
            if p.S is not None and t == self.ts:
                bt += gg[0, 1]
            bt += self.alpha * (q[:, t] - q[:, t - 1])
            if t < T - 1:
                bt -= self.alpha * (q[:, t + 1] - q[:, t])
            g[:, t] = 2.0 * bt
        return g[:, 2:].reshape(-1)

    def grad_fd(self, x, h=1e-6):
        E = np.eye(len(x)) * h
        return (self.f(x[None] + E) - self.f(x[None] - E)) / (2.0 * h)

    def projected_gradient(self, x, tol=1e-12):
        g = self.grad(x)
        g[(x <= self.lo_x + tol) & (g > 0)] = 0.0
        g[(x >= self.hi_x - tol) & (g < 0)] = 0.0
        return g

    def lbfgsb(self, x0):
        from scipy.optimize import minimize
        bounds = list(zip(self.lo_x, self.hi_x))
        return minimize(lambda x: float(self.f(x)[0]), np.clip(x0, self.lo_x, self.hi_x), jac=self.grad, method="L-BFGS-B", bounds=bounds,
                        options=dict(maxiter=6000, maxfun=30000, ftol=1e-16, gtol=1e-11))


def check_against_lbfgsb(oracle_obj, prob, opts, Qsol, cost, status, min_same_basin):
    """Every solution must be a stationary point of the bound-constrained problem; where L-BFGS-B converges into the
    same basin (|dQ| < 1e-3 rad), minimiser and minimum must agree.  Returns the number of same-basin instances."""
    same = 0
    for b in range(prob.B):
        sp = SmoothProblem(oracle_obj, prob, b, opts, Qsol[b])
        xs = sp.pack(Qsol[b])
        assert status[b] == 0
        np.testing.assert_allclose(sp.f(xs)[0], cost[b], rtol=1e-11)             # the reported cost is the objective at the reported point
        assert np.abs(sp.projected_gradient(xs)).max() < 1e-6                    # ... which is a KKT point of the bound-constrained problem
        assert cost[b] <= sp.f(np.clip(sp.pack(prob.Q0[b]), sp.lo_x, sp.hi_x))[0] * (1 + 1e-12)  # never worse than the (clipped) seed
        res = sp.lbfgsb(sp.pack(prob.Q0[b]))
        if np.abs(sp.projected_gradient(res.x)).max() < 1e-6 and np.abs(res.x - xs).max() < 1e-3:
            same += 1
            np.testing.assert_allclose(res.fun, cost[b], rtol=1e-9)
            assert np.abs(res.x - xs).max() < 2e-5
    assert same >= min_same_basin, f"only {same} of {prob.B} instances ended in L-BFGS-B's basin"
    return same


def lbfgsb_fd(fbatch, x0, lo, hi, h=1e-6):
    """L-BFGS-B on a black-box objective given as a batch function fbatch(X [k, n]) -> [k]; central finite differences."""
    from scipy.optimize import minimize
    x0 = np.clip(np.asarray(x0, dtype=np.float64), lo, hi)
    E = np.eye(len(x0)) * h

    def fg(x):
        v = fbatch(np.concatenate([x[None], x[None] + E, x[None] - E]))
        n = len(x)
        return float(v[0]), (v[1:n + 1] - v[n + 1:]) / (2.0 * h)

    return minimize(fg, x0, jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)), options=dict(maxiter=4000, maxfun=20000, ftol=1e-16, gtol=1e-10))


def check_ik_against_lbfgsb(oracle_obj, prob, q0, q_sol, cost, min_agree):
    """IK without the collision term (gto/ik_solver.py:30-76 pose matching under the joint limits): the solver's answer
    against L-BFGS-B on the objective value alone (the oracle evaluates it with max_iter = 0).  A 7-joint arm reaches a
    pose along a one-parameter family of configurations, so minimisers are compared through their objective values:
    both solve the pose (f ~ 0), or both stop at the same positive minimum, or the solver's is the lower one; and
    L-BFGS-B restarted at the solver's answer finds nothing lower."""
    d = prob.desc
    oi, lo, hi = d.opt_index, d.lower[d.opt_index], d.upper[d.opt_index]
    agree = 0
    for b in range(len(q0)):
        def fbatch(X, b=b):
            q = np.repeat(q_sol[b:b + 1], len(X), 0)
            q[:, oi] = X
            return oracle_obj.solve_ik_batch(None, q, np.repeat(prob.goals[b:b + 1, 0], len(X), 0), None, max_iter=0, n_threads=1)[1]
        np.testing.assert_allclose(fbatch(q_sol[b:b + 1, oi])[0], cost[b], rtol=1e-9, atol=1e-14)
        res = lbfgsb_fd(fbatch, q0[b, oi], lo, hi)
        if (cost[b] < 1e-12 and res.fun < 1e-12) or abs(res.fun - cost[b]) <= 1e-7 * abs(cost[b]):
            agree += 1
        else:
            assert cost[b] < res.fun  # a different basin: then the solver's must be the better one
        r2 = lbfgsb_fd(fbatch, q_sol[b, oi], lo, hi)
        assert r2.fun >= cost[b] * (1 - 1e-7) - 1e-12
    assert agree >= min_agree, f"only {agree} of {len(q0)} IK instances agree with L-BFGS-B"


def check_base_against_lbfgsb(oracle_obj, desc, qc, goals, w, y_sol, q_sol, cost, min_agree):
    """Base placement (gto/base_planner.py:35-123) with a firm effort weight: unknowns (x, y, theta) + one arm
    configuration per goal, against L-BFGS-B on the objective value alone.  The effort term makes the base pose unique;
    the 7-joint arm still reaches each goal along a family of configurations, so what is compared is the minimum and the
    base pose, not the arm angles."""
    oi, lo, hi = desc.opt_index, desc.lower[desc.opt_index], desc.upper[desc.opt_index]
    B, n_goals, n = goals.shape[0], goals.shape[1], len(oi)
    lo_x = np.concatenate([[-np.inf, -np.inf, -np.pi], np.tile(lo, n_goals)])
    hi_x = np.concatenate([[np.inf, np.inf, np.pi], np.tile(hi, n_goals)])
    agree = 0
    for b in range(B):
        def fbatch(X, b=b):
            k = len(X)
            q = np.repeat(q_sol[b:b + 1], k, 0)
            q[:, :, oi] = X[:, 3:].reshape(k, n_goals, n)
            return oracle_obj.eval_base_objective(X[:, :3], q, np.repeat(goals[b:b + 1], k, 0), None, w)
        xs = np.concatenate([y_sol[b], q_sol[b][:, oi].reshape(-1)])
        np.testing.assert_allclose(fbatch(xs[None])[0], cost[b], rtol=1e-9, atol=1e-13)
        x0 = np.concatenate([np.zeros(3), np.tile(qc[b, oi], n_goals)])
        res = lbfgsb_fd(fbatch, x0, lo_x, hi_x)
        if abs(res.fun - cost[b]) <= 1e-7 * max(abs(cost[b]), 1e-6) and np.abs(res.x[:3] - y_sol[b]).max() < 1e-5:
            agree += 1
        else:
            assert cost[b] < res.fun
        r2 = lbfgsb_fd(fbatch, xs, lo_x, hi_x)
        assert r2.fun >= cost[b] * (1 - 1e-7) - 1e-12
    assert agree >= min_agree, f"only {agree} of {B} base placements agree with L-BFGS-B"


def check_obstacle_blocks_against_fd(obj, desc, T, ts, Q, base, h=1e-6):
    """The obstacle term's Gauss-Newton blocks assembled point by point in numpy — cost value and field-gradient model
    per surface point from eval_points (pinned against gto/sdf_callback.py), point Jacobians dx/dq by central differences
    of the world points — against eval_obstacle_normal_eq (wrench Grams folded per link and projected onto the joint
    screws).  `obj` is the oracle or the HIP handle: same surface.  Returns the number of waypoints with a non-zero block."""
    oi = desc.opt_index
    n = len(oi)
    JtJ, Jtr, ss = obj.eval_obstacle_normal_eq(0, base, Q[None])
    nonzero = 0
    for t in range(T):
        q = Q[:, t]
        use_obs = t >= ts  # gto/gto_planner.py:117-131: sdf_cost_all before the standoff waypoint, sdf_cost_obstacle from it on
        _, _, val, grad = obj.eval_points(0, q[None], base, use_obs=use_obs)
        np.testing.assert_allclose(ss[0, t], (val[0] ** 2).sum(), rtol=1e-12, atol=1e-14)
        if t < 2:
            continue  # pinned waypoints: no unknowns
        qs = np.repeat(q[None], 2 * n, 0)
        for j in range(n):
            qs[2 * j, oi[j]] += h
            qs[2 * j + 1, oi[j]] -= h
        xp = obj.eval_points(0, qs, base, use_obs=use_obs)[0]
        dx = (xp[0::2] - xp[1::2]) / (2.0 * h)              # [n, P, 3]
        J = np.einsum("pk,jpk->pj", grad[0], dx)            # [P, n] rows of the obstacle Jacobian
        A, b = J.T @ J, J.T @ val[0]
        scale = max(np.abs(A).max(), 1e-12)
        np.testing.assert_allclose(JtJ[0, t], A, rtol=0, atol=2e-6 * scale)
        np.testing.assert_allclose(Jtr[0, t], b, rtol=0, atol=2e-6 * max(np.abs(b).max(), 1e-12))
        nonzero += int(np.abs(A).max() > 0)
    return nonzero
