"""OpTaS-shaped problem/solver surface for the GTO path, without CasADi.

The reference builds the trajectory NLP symbolically (optas.OptimizationBuilder -> CasADi graph,
optas/builder.py:12-636) and hands it to IPOPT through optas.CasADiSolver (optas/solver.py:323-421).
Here the builder *records* the same problem structurally (named decision variables, parameters,
structured cost terms, linear constraints) and the solver object drives the MI355X solver behind
the C ABI (include/gto_solver.h).  Method names, argument meaning, dictionary keys, array shapes
and the "return the iterate even if not converged" behaviour (optas/solver.py:64,135) are kept, so
code written against the reference's GTOPlanner call pattern (gto/gto_planner.py:44-245) works.

Only the cost structure of the GTO path is representable (that is the scope of this package):
    GoalSetPointMatching + ObstacleField + JointVelocity, initial configuration, Euler dynamics,
    joint limits.  Anything else raises NotImplementedError instead of being silently ignored.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np


class DM(np.ndarray):
    """Minimal stand-in for casadi.DM results: an ndarray with ``toarray()`` (gto/gto_planner.py:245)."""

    def __new__(cls, a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            a = a.reshape(-1, 1)  # casadi vectors are columns
        return a.view(cls)

    def toarray(self) -> np.ndarray:
        return np.asarray(self)


# --------------------------------------------------------------------------------- structured cost terms
@dataclass
class GoalSetPointMatching:
    """min over goals of the gripper point-matching cost (+ standoff), gto/gto_planner.py:84-105."""
    link_ee: str
    link_gripper: str
    goal_size: int
    use_standoff: bool
    standoff_pose: Optional[np.ndarray]  # 4x4, optas/spatialmath.py:160-183
    standoff_offset: int


@dataclass
class ObstacleField:
    """weight * (sum c_all[off]^2 before the standoff waypoint + sum c_obs[off]^2 after), :108-131."""
    weight: float
    standoff_offset: int


@dataclass
class JointVelocity:
    """weight * sumsqr(dQ), gto/gto_planner.py:133-135."""
    weight: float


@dataclass
class GTOProblem:
    """What OptimizationBuilder.build() returns: the recorded structure (cf. optas/optimization.py:447-486)."""
    T: int
    robot: object
    decision_variables: "OrderedDict[str, tuple]"
    parameters: "OrderedDict[str, tuple]"
    cost_terms: "OrderedDict[str, object]"
    has_initial_configuration: bool
    has_initial_velocity_zero: bool
    dt: Optional[float]
    enforce_limits: bool

    @property
    def nx(self) -> int:
        return int(sum(a * b for a, b in self.decision_variables.values()))

    @property
    def np(self) -> int:
        return int(sum(a * b for a, b in self.parameters.values()))


class OptimizationBuilder:
    """Records the GTO problem with the reference builder's method names (optas/builder.py)."""

    def __init__(self, T: int, robots=None, tasks=None, derivs_align: bool = False):
        assert T > 0, "T must be strictly positive"
        robots = robots if isinstance(robots, list) else [robots]
        if tasks:
            raise NotImplementedError("task models are outside the GTO path (SURVEY.md section 2, row 13)")
        if len(robots) != 1:
            raise NotImplementedError("exactly one robot model is supported")
        self.T = T
        self.robot = robots[0]
        self._vars: "OrderedDict[str, tuple]" = OrderedDict()
        self._params: "OrderedDict[str, tuple]" = OrderedDict()
        self._costs: "OrderedDict[str, object]" = OrderedDict()
        self._init_q = False
        self._init_dq = False
        self._dt: Optional[float] = None
        self._limits = False
        name = self.robot.get_name()
        n_opt, n_par = self.robot.num_opt_joints, self.robot.num_param_joints
        for d in self.robot.time_derivs:  # optas/builder.py:90-100: t = T - d
            prefix = name + "/" + "d" * d + "q"
            self._vars[prefix + "/x"] = (n_opt, T - d)
            self._params[prefix + "/p"] = (n_par, T - d)

    # ---- the calls GTOPlanner.setup_optimization makes (gto/gto_planner.py:44-142)
    def add_parameter(self, name: str, m: int = 1, n: int = 1):
        if name in self._params:
            raise KeyError(f"'{name}' already exists")
        self._params[name] = (int(m), int(n))
        return name

    def add_decision_variables(self, name: str, m: int = 1, n: int = 1, is_discrete: bool = False):
        raise NotImplementedError("extra decision variables are outside the GTO path")

    def get_model_names(self) -> List[str]:
        return [self.robot.get_name()]

    def initial_configuration(self, name: str, init=None, time_deriv: int = 0):
        if time_deriv == 0:
            self._init_q = True
        elif time_deriv == 1:
            if init is not None and np.any(np.asarray(init) != 0):
                raise NotImplementedError("only a zero initial velocity is supported (gto/gto_planner.py:63-65)")
            self._init_dq = True
        else:
            raise NotImplementedError("time_deriv > 1")

    def integrate_model_states(self, name: str, time_deriv: int, dt: float):
        if time_deriv != 1:
            raise NotImplementedError("only velocity -> position integration (gto/gto_planner.py:68-72)")
        self._dt = float(dt)

    def get_robot_states_and_parameters(self, name: str, time_deriv: int = 0):
        return name + "/" + "d" * time_deriv + "q"

    get_model_states = get_robot_states_and_parameters

    def add_cost_term(self, name: str, cost_term):
        if not isinstance(cost_term, (GoalSetPointMatching, ObstacleField, JointVelocity)):
            raise NotImplementedError(
                "only the GTO cost structure is representable: pass GoalSetPointMatching / ObstacleField / "
                "JointVelocity (gto/gto_planner.py:84-135); symbolic CasADi expressions are not supported")
        if name in self._costs:
            raise KeyError(f"'{name}' already exists")
        self._costs[name] = cost_term

    def enforce_model_limits(self, name: str, time_deriv: int = 0, lower=None, upper=None):
        if time_deriv != 0 or lower is not None or upper is not None:
            raise NotImplementedError("only URDF position limits (gto/gto_planner.py:138)")
        self._limits = True

    def add_equality_constraint(self, *a, **k):
        raise NotImplementedError("general constraints are outside the GTO path")

    add_leq_inequality_constraint = add_geq_inequality_constraint = add_bound_inequality_constraint = add_equality_constraint

    def build(self) -> GTOProblem:
        return GTOProblem(T=self.T, robot=self.robot, decision_variables=self._vars, parameters=self._params,
                          cost_terms=self._costs, has_initial_configuration=self._init_q,
                          has_initial_velocity_zero=self._init_dq, dt=self._dt, enforce_limits=self._limits)


# --------------------------------------------------------------------------------- solver
class Solver:
    """optas.Solver contract (optas/solver.py:61-316)."""

    def __init__(self, optimization: GTOProblem, error_on_fail: bool = False):
        self.opt = optimization
        self._error_on_fail = error_on_fail
        self.x0: Dict[str, np.ndarray] = {}
        self._p_dict: Dict[str, np.ndarray] = {}
        self._stats: Dict = {}

    def reset_initial_seed(self, x0: Dict[str, np.ndarray]) -> None:
        for k in x0:
            if k not in self.opt.decision_variables:
                raise KeyError(f"unknown decision variable '{k}'")
        # missing entries default to zeros (optas/mx_container.py:113-123)
        self.x0 = {k: np.zeros(shape) for k, shape in self.opt.decision_variables.items()}
        for k, v in x0.items():
            self.x0[k] = np.asarray(v, dtype=np.float64).reshape(self.opt.decision_variables[k])

    def reset_parameters(self, p: Dict[str, np.ndarray]) -> None:
        for k in p:
            if k not in self.opt.parameters:
                raise KeyError(f"unknown parameter '{k}'")
        self._p_dict = {k: np.zeros(shape, dtype=np.float32 if k in ("sdf_cost_all", "sdf_cost_obstacle") else np.float64)
                        for k, shape in self.opt.parameters.items() if k not in p}
        from .depth_scene import LazyCostField
        for k, v in p.items():
            if isinstance(v, LazyCostField):  # a cost field that is resident on the device (depth_scene.py): stays there
                self._p_dict[k] = v
                continue
            # the two cost fields stay float32 (they are float32 at the boundary, include/gto_solver.h): converting 2 x 2 M
            # voxels to float64 here and back in set_scene cost more than the upload itself
            keep = k in ("sdf_cost_all", "sdf_cost_obstacle") and getattr(v, "dtype", None) == np.float32
            self._p_dict[k] = (np.asarray(v) if keep else np.asarray(v, dtype=np.float64)).reshape(self.opt.parameters[k])
        self._scene_dirty = True

    def stats(self) -> Dict:
        return self._stats

    def did_solve(self) -> bool:
        return bool(self._stats.get("success", False))

    def number_of_iterations(self) -> int:
        return int(self._stats.get("iter_count", 0))


class CasADiSolver(Solver):
    """Same call pattern as optas.CasADiSolver (``CasADiSolver(builder.build()).setup("ipopt", ...)``,
    gto/gto_planner.py:141-142) backed by the MI355X Gauss-Newton/LM solver.  ``solver_name`` is
    accepted for compatibility; IPOPT's ``max_iter`` maps onto the iteration cap."""

    def setup(self, solver_name: str = "ipopt", solver_options: Optional[Dict] = None):
        from . import _capi
        prob = self.opt
        costs = prob.cost_terms
        goal = next((c for c in costs.values() if isinstance(c, GoalSetPointMatching)), None)
        obst = next((c for c in costs.values() if isinstance(c, ObstacleField)), None)
        vel = next((c for c in costs.values() if isinstance(c, JointVelocity)), None)
        if goal is None or vel is None:
            raise NotImplementedError("the GTO problem needs a goal-set term and a joint-velocity term")
        if not (prob.has_initial_configuration and prob.has_initial_velocity_zero and prob.dt and prob.enforce_limits):
            raise NotImplementedError("the GTO problem fixes the initial state, integrates velocities and "
                                      "enforces joint limits (gto/gto_planner.py:58-72,138)")
        opts = _capi.default_opts()
        opts.T = prob.T
        opts.Tmax = prob.dt * (prob.T - 1)
        opts.standoff_offset = goal.standoff_offset
        opts.w_obstacle = obst.weight if obst is not None else 0.0
        opts.w_vel = vel.weight
        so = (solver_options or {}).get(solver_name, solver_options or {})
        if "max_iter" in so:
            opts.max_iter = int(so["max_iter"])
        self._goal = goal
        self._handle = prob.robot.solver_handle(goal.link_ee, goal.link_gripper, opts, role="planner")
        self._handle.set_opts(max_iter=opts.max_iter, w_obstacle=opts.w_obstacle, w_vel=opts.w_vel)
        self._name = prob.robot.get_name()
        return self

    SCENE_ID = 0

    def ensure_scene(self) -> int:
        """Upload the cost fields of the current parameters (once per reset_parameters) and return the scene id:
        seed scoring (gto/gto_planner.py:208) and the solve read the same resident scene."""
        # the handle (and its scene 0) is shared by every solver with the same role / links / T on this robot model: another
        # planner's upload replaces the field under this solver, so "already uploaded" is only true while this solver was the
        # last one to write the scene
        from .depth_scene import resident_of
        p = self._p_dict
        la, lo = p.get("sdf_cost_all"), p.get("sdf_cost_obstacle")
        shared = getattr(self, "_shared_from", None)  # (handle, scene id, generation) of the resident scene this solver borrows
        stale = shared is not None and shared[0].scene_generation(shared[1]) != shared[2]  # rebuilt since: the borrowed pointers are gone
        if getattr(self, "_scene_dirty", True) or stale or getattr(self._handle, "_scene0_owner", None) is not self:
            robot = self.opt.robot
            ro, ra = resident_of(lo), resident_of(la)  # the obstacle field first: its build holds both fields
            if ro is not None and ra is not None and ra[0] is ro[0] and (ra.sid, ra.gen) == (ro.sid, ro.gen):
                # both fields live in ONE build of one resident scene: the solver's scene becomes that scene, each field the
                # half it is (no copy, nothing through the host)
                self._handle.share_scene(self.SCENE_ID, ro.handle, ro.sid, all_from=ra.half, obs_from=ro.half)
                self._shared_from = (ro.handle, ro.sid, ro.gen)
            else:
                shape, origin, res = robot.field_geometry()
                # parameters that were never set are zeros (optas/mx_container.py:121): plan() leaves sdf_cost_all out
                c_all = p.get("sdf_cost_all", np.zeros(int(np.prod(shape))))
                c_obs = p.get("sdf_cost_obstacle", np.zeros(int(np.prod(shape))))
                self._handle.set_scene(self.SCENE_ID, np.asarray(c_all), np.asarray(c_obs), shape, origin, res)
                self._shared_from = None
            self._handle._scene0_owner = self
            self._scene_dirty = False
        return self.SCENE_ID

    def solve(self) -> Dict[str, DM]:
        robot, T, name = self.opt.robot, self.opt.T, self._name
        p = self._p_dict
        ndof = robot.ndof
        oi, pi = robot.optimized_joint_indexes, robot.parameter_joint_indexes
        Q0 = np.zeros((ndof, T))
        Q0[oi] = self.x0.get(f"{name}/q/x", np.zeros((len(oi), T)))
        if len(pi):
            Q0[pi] = p[f"{name}/q/p"]
        n = self._goal.goal_size
        # tf_goal column i = RT_i.flatten() row-major (gto/gto_planner.py:188-191)
        goals = p["tf_goal"].T.reshape(1, n, 16)
        sid = self.ensure_scene()
        S = self._goal.standoff_pose if self._goal.use_standoff else None
        Q, dQ, cost, iters, status = self._handle.solve_batch(
            sid, p["qc"].reshape(1, ndof), goals, n, S, p["base_position"].reshape(1, 3), Q0[None])
        self._stats = {"iter_count": int(iters[0]), "success": bool(status[0] == 0),
                       "return_status": ("Solve_Succeeded", "Maximum_Iterations_Exceeded", "Numerical_Failure")[int(status[0])]}
        if self._error_on_fail and not self.did_solve():
            raise RuntimeError("Solver failed!")
        sol = {f"{name}/q/x": DM(Q[0][oi]), f"{name}/dq/x": DM(dQ[0][oi]), "f": DM(cost.reshape(1, 1))}
        sol[f"{name}/q"] = DM(Q[0])    # parameter joints re-inserted (optas/solver.py:139-157)
        sol[f"{name}/dq"] = DM(dQ[0])
        return sol
