#!/usr/bin/env python3
"""Pin the ASSEMBLED objective: execute the reference's own problem-definition code numerically.

Build-container only (reads /root/reference); writes tests/golden/objective.npz (data only).
Re-run:  python tests/golden/make_objective_golden.py

What executes from the reference (class bodies extracted with ast and exec'd, because the modules import
casadi / trimesh / pyrender / transforms3d at import time and cannot be imported here):
  gto/gto_planner.py   GTOPlanner.__init__, setup_optimization (:42-142: goal-set + standoff cost incl. the
                       tf_goal[:, i].reshape((4, 4)).T decode, obstacle split at T + standoff_offset with sumsqr,
                       velocity cost, the constraint calls), plan (:145-182), plan_goalset (:185-245: seed
                       construction, scoring with compute_plan_cost, lexsort, interpolate=False seeds, the
                       parameter dictionaries)
  gto/ik_solver.py     IKSolver.__init__, setup_optimization (:30-76)
  gto/base_planner.py  BasePlanner.__init__, setup_optimization (:35-93)
  gto/gto_models.py    points_to_offsets (:174-187), points_to_offsets_numpy, compute_plan_cost,
                       compute_fk_surface_points (function bodies)
  gto/utils.py         interpolate_waypoints;  optas/spatialmath.py standoff, rotz, rt2tr, rpy2r, invt;
  optas/models.py      RobotModel.get_global_link_transform / get_link_transform (FK), TaskModel
The CasADi layer is replaced by NUMERIC stand-ins with CasADi's semantics: the "symbolic" arrays the builder
hands out are numpy arrays holding concrete values, reshape is column-major as in CasADi, a parametric gather
`field[offsets]` indexes with the float-valued offsets, optas.sumsqr / mmin / sum1 / floor / horzcat are the
obvious numpy functions.  Every add_cost_term() call of the reference is then a NUMBER, recorded by name.
The robot's surface points (unseeded random samples in the reference, gto/gto_models.py:76-77) are an input:
they come from this repo's distilled descriptions (grasptrajopt_amd/data/*.npz).
"""
import ast
import collections
import contextlib
import io
import os
import sys
import textwrap
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _reference_stubs as stubs  # noqa: E402

REF = stubs.REF


# ------------------------------------------------------------------------------- numeric CasADi stand-ins
class Num(np.ndarray):
    """Concrete values where the reference handles MX expressions.  CasADi semantics that differ from numpy:
    reshape is column-major; indexing with a float-valued index array gathers (MX[MX] parametric get_nz)."""

    def __new__(cls, a):
        return np.array(a, dtype=np.float64).view(cls)

    def reshape(self, *shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return np.asarray(self).reshape(shape, order="F").view(Num)

    def __getitem__(self, idx):
        if isinstance(idx, np.ndarray) and idx.dtype.kind == "f":
            assert np.all(idx == np.round(idx))
            idx = np.asarray(idx).astype(np.int64)
        out = np.asarray(self)[idx]
        return out.view(Num) if isinstance(out, np.ndarray) else out

    def toarray(self):
        return np.asarray(self)


class Col(Num):
    """Column vector whose integer index gives a 1x1 array, as a CasADi DM does."""

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return Num(np.asarray(self).reshape(-1)[idx].reshape(1, 1))
        return Num.__getitem__(self, idx)


def extract_class(path, name, ns):
    src = open(path).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ClassDef) and node.name == name:
            code = textwrap.dedent(ast.get_source_segment(src, node))
            exec(compile(code, path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


def extract_functions(path, names, ns):
    src = open(path).read()
    out = {}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = textwrap.dedent(ast.get_source_segment(src, node))
            exec(compile(code, path, "exec"), ns)
            out[node.name] = ns[node.name]
    return out


class Ctx:
    """Values the fake builder hands out."""
    values = {}
    last = None  # the builder of the most recent setup_optimization


class FakeBuilder:
    def __init__(self, T, robots=(), tasks=()):
        self.T, self.robots, self.tasks = T, list(robots), list(tasks)
        self.costs = collections.OrderedDict()
        self.mmin_args = []
        self.calls = []
        Ctx.last = self

    def add_parameter(self, name, m=1, n=1):
        v = np.asarray(Ctx.values[name], dtype=np.float64)
        if n == 1 and v.ndim == 1:
            assert v.shape[0] == m, (name, v.shape, m)
        else:
            assert v.shape == (m, n), (name, v.shape, (m, n))
        return Num(v)

    def get_robot_states_and_parameters(self, name, time_deriv=0):
        return Num(Ctx.values[f"{name}/{'d' * time_deriv}q"])

    def get_model_states(self, name, time_deriv=0):
        return Num(Ctx.values[f"{name}/y"])

    def add_cost_term(self, name, value):
        self.costs[name] = float(np.asarray(value).reshape(-1)[0])

    def initial_configuration(self, name, init=None, time_deriv=0):
        self.calls.append(("initial_configuration", time_deriv, None if init is None else np.asarray(init).ravel().copy()))

    def integrate_model_states(self, name, time_deriv, dt):
        self.calls.append(("integrate_model_states", time_deriv, float(dt)))

    def enforce_model_limits(self, name, time_deriv=0):
        self.calls.append(("enforce_model_limits", time_deriv, None))

    def add_bound_inequality_constraint(self, name, lo, x, hi):
        self.calls.append(("bound", name, (float(lo), float(hi))))

    def build(self):
        return self


class FakeSolver:
    """Records what the planner hands to the solver; solve() returns the seed as 'solution'."""
    last = None

    def __init__(self, optimization):
        self.opt = optimization
        FakeSolver.last = self

    def setup(self, name, solver_options=None):
        self.name, self.options = name, solver_options
        return self

    def reset_initial_seed(self, d):
        self.seed = {k: np.array(v, dtype=np.float64) for k, v in d.items()}

    def reset_parameters(self, d):
        self.params = {k: np.array(v, dtype=np.float64) for k, v in d.items()}

    def solve(self):
        rn = self.opt.robots[0].get_name()
        robot = self.opt.robots[0]
        ndof, T = robot.ndof, self.opt.T
        Q = np.zeros((ndof, T))
        Q[robot.optimized_joint_indexes, :] = self.seed[f"{rn}/q/x"]
        Q[robot.parameter_joint_indexes, :] = self.params[f"{rn}/q/p"]
        return {f"{rn}/q": Num(Q), f"{rn}/dq": Num(np.zeros((ndof, T - 1))), "f": Num(np.zeros((1, 1)))}


def _mmin(b):
    return lambda cost: (b().mmin_args.append(np.array(cost, dtype=np.float64).ravel().copy()), float(np.min(cost)))[1]


def make_namespaces(ref):
    optas = types.SimpleNamespace(
        OptimizationBuilder=FakeBuilder, CasADiSolver=FakeSolver,
        sumsqr=lambda a: float(np.sum(np.asarray(a) ** 2)), sum1=lambda a: float(np.sum(np.asarray(a))),
        mmin=_mmin(lambda: Ctx.last), floor=lambda a: Num(np.floor(np.asarray(a))),
        horzcat=lambda *a: Num(np.hstack([np.asarray(x) for x in a])),
        linspace=lambda a, b, n: np.linspace(a, b, n).reshape(-1, 1).view(Col),
        DM=Num, diag=lambda v: np.diag(np.asarray(v, dtype=np.float64).ravel()))
    optas.DM.ones = staticmethod(lambda n, m=1: Num(np.ones((n, m))))
    cs = types.SimpleNamespace(
        MX=types.SimpleNamespace(zeros=lambda n: Num(np.zeros(n))),
        fmax=lambda a, b: np.maximum(a, b), fmin=lambda a, b: np.minimum(a, b),
        horzcat=lambda *a: np.array([float(x) for x in a]))
    sm = ref.spatialmath
    ns = dict(np=np, optas=optas, cs=cs, standoff=sm.standoff, rotz=sm.rotz, rt2tr=sm.rt2tr,
              interpolate_waypoints=ref.utils.interpolate_waypoints, TaskModel=ref.models.TaskModel, print=lambda *a, **k: None)
    return ns


class FakeRobot:
    """What GTORobotModel offers the planner, with FK executed by the reference's RobotModel."""

    def __init__(self, ref, robot, desc_npz, desc_json, grid, ns):
        cfg = yaml.safe_load(open(f"{REF}/data/configs/{robot}.yaml"))["robot_cfg"]
        self.cfg = cfg
        self.m = ref.models.RobotModel(urdf_filename=f"{REF}/{cfg['urdf_robot_path']}", time_derivs=[0, 1],
                                       param_joints=cfg["param_joints"])
        self.ndof = self.m.ndof
        self.optimized_joint_indexes = list(self.m.optimized_joint_indexes)
        self.parameter_joint_indexes = list(self.m.parameter_joint_indexes)
        urdf = self.m.get_urdf()
        sm = ref.spatialmath
        names = [l.name for l in urdf.links if l.visual is not None and l.name in cfg["collision_link_names"]]
        assert names == list(desc_json["link_names"]), (names, desc_json["link_names"])
        self.surface_pc_map = collections.OrderedDict()
        self.visual_tf = {}
        for li, name in enumerate(names):
            pts = desc_npz["points"][desc_npz["point_link"] == li]
            self.surface_pc_map[name] = types.SimpleNamespace(points=pts, normals=np.zeros_like(pts))
            xyz, rpy = self.m.get_link_visual_origin(urdf.link_map[name])
            V = np.asarray(sm.rt2tr(sm.rpy2r(rpy), xyz))  # gto/gto_models.py:95-96
            self.visual_tf[name] = (lambda nm, Vv: lambda q: Num(np.asarray(self.m.get_global_link_transform(nm, np.asarray(q).ravel())) @ Vv))(name, V)
        # grid geometry as setup_points_field leaves it (gto/gto_models.py:155-171)
        self.origin = np.asarray(grid["origin"], dtype=np.float64).reshape(1, 3)
        self.grid_resolution = float(grid["res"])
        self.field_shape = tuple(int(x) for x in grid["shape"])
        self.field_size = int(np.prod(self.field_shape))
        fns = extract_functions(f"{REF}/gto/gto_models.py",
                                {"points_to_offsets", "points_to_offsets_numpy", "compute_plan_cost", "compute_fk_surface_points"}, ns)
        for k, f in fns.items():
            setattr(self, k, types.MethodType(f, self))

    def get_name(self):
        return self.m.get_name()

    def extract_optimized_dimensions(self, v):
        v = np.asarray(v)
        return Num((v.reshape(-1, 1) if v.ndim == 1 else v)[self.optimized_joint_indexes, :])

    def extract_parameter_dimensions(self, v):
        v = np.asarray(v)
        return Num((v.reshape(-1, 1) if v.ndim == 1 else v)[self.parameter_joint_indexes, :])

    def get_global_link_transform_function(self, link, n=1):
        one = lambda q: Num(np.asarray(self.m.get_global_link_transform(link, np.asarray(q).ravel())))
        if n > 1:  # ListFunction of optas/models.py:741-751
            return lambda Q: [one(np.asarray(Q)[:, i]) for i in range(np.asarray(Q).shape[1])]
        return lambda Q: one(np.asarray(Q).reshape(self.ndof, -1)[:, 0])

    def get_link_transform_function(self, link, base_link):
        return lambda q: Num(np.asarray(self.m.get_link_transform(link, np.asarray(q).ravel(), base_link)))


def random_field(rng, shape, sparse):
    f = rng.random(shape).astype(np.float32)
    if sparse:  # mostly free space, like a real cost field
        f[rng.random(shape) < 0.85] = 0.0
    return f.reshape(-1)


def smooth_traj(rng, qc, lo, hi, opt, T):
    """In-limit trajectory (ndof, T): starts at qc (first two waypoints pinned), wanders to a random configuration."""
    ndof = len(qc)
    qg = qc.copy()
    qg[opt] = rng.uniform(lo[opt], hi[opt])
    s = np.linspace(0, 1, T)
    Q = qc[:, None] + (qg - qc)[:, None] * (s * s * (3 - 2 * s))[None, :]
    Q[opt] += 0.05 * rng.standard_normal((len(opt), T)) * np.sin(np.pi * s)[None, :]
    Q[opt] = np.clip(Q[opt], lo[opt][:, None], hi[opt][:, None])
    Q[:, 1] = Q[:, 0] = qc
    return Q


def run_robot(ref, robot, rng, out):
    import json
    dz = np.load(f"{ROOT}/grasptrajopt_amd/data/{robot}.npz")
    dj = json.load(open(f"{ROOT}/grasptrajopt_amd/data/{robot}.json"))
    # a non-cubic grid at the reference's 5 cm resolution around the arm (shape order matters: x slowest)
    grid = dict(origin=np.array([-0.62, -0.93, -0.31]) if robot == "panda" else np.array([-0.45, -1.02, 0.02]),
                res=0.05, shape=(34, 40, 30))
    ns = make_namespaces(ref)
    R = FakeRobot(ref, robot, dz, dj, grid, ns)
    cfg = R.cfg
    T = 50
    link_ee, link_gr = cfg["link_ee"], cfg["link_gripper"]
    qc = np.array(cfg["default_pose"], dtype=np.float64)
    lo = np.maximum(np.asarray(R.m.lower_actuated_joint_limits).ravel(), -3.0)
    hi = np.minimum(np.asarray(R.m.upper_actuated_joint_limits).ravel(), 3.0)
    opt, par = np.array(R.optimized_joint_indexes), np.array(R.parameter_joint_indexes)
    GTOPlanner = extract_class(f"{REF}/gto/gto_planner.py", "GTOPlanner", ns)
    IKSolver = extract_class(f"{REF}/gto/ik_solver.py", "IKSolver", ns)
    BasePlanner = extract_class(f"{REF}/gto/base_planner.py", "BasePlanner", ns)
    rn = R.get_name()
    fields = dict(dense_all=random_field(rng, grid["shape"], False), dense_obs=random_field(rng, grid["shape"], False),
                  sparse_all=random_field(rng, grid["shape"], True), sparse_obs=random_field(rng, grid["shape"], True))
    out[f"{robot}_grid_origin"], out[f"{robot}_grid_shape"], out[f"{robot}_grid_res"] = grid["origin"], np.array(grid["shape"]), grid["res"]
    for k, v in fields.items():
        out[f"{robot}_field_{k}"] = v
    out[f"{robot}_points_checksum"] = np.array([dz["points"].sum(), np.abs(dz["points"]).sum()])
    out[f"{robot}_qc"] = qc
    axis = cfg["axis_standoff"]

    # ---------------- trajectory objective: f_goal per goal, f_obs, f_vel (gto/gto_planner.py:84-135)
    cases = []
    nQ = 8
    Qs = np.stack([smooth_traj(rng, qc, lo, hi, opt, T) for _ in range(nQ)])
    dt = 10.0 / 49
    for n_goals in (1, 4):
        for use_so in (False, True):
            for field_kind in ("dense", "sparse"):
                planner = GTOPlanner(R, link_ee, link_gr, standoff_distance=-0.1, standoff_offset=-10)
                # goals: end-effector poses of random in-limit configurations
                qg = rng.uniform(lo, hi, size=(nQ, n_goals, R.ndof))
                RT = np.array([[np.asarray(R.m.get_global_link_transform(link_ee, qg[i, g])) for g in range(n_goals)] for i in range(nQ)])
                base = rng.uniform(-0.08, 0.08, size=(nQ, 3)) * (1 if field_kind == "dense" else 0)
                fg_all, fo, fv = np.zeros((nQ, n_goals)), np.zeros(nQ), np.zeros(nQ)
                for i in range(nQ):
                    Q = Qs[i]
                    dQ = np.zeros((R.ndof, T - 1))
                    dQ[opt] = (Q[opt, 1:] - Q[opt, :-1]) / dt  # what the dynamics constraint makes of Q (gto/gto_planner.py:68-72)
                    tf_goal = np.stack([RT[i, g].flatten() for g in range(n_goals)], axis=1)  # :188-191
                    Ctx.values = {"qc": qc, "tf_goal": tf_goal, "sdf_cost_all": fields[f"{field_kind}_all"],
                                  "sdf_cost_obstacle": fields[f"{field_kind}_obs"], "base_position": base[i],
                                  f"{rn}/q": Q, f"{rn}/dq": dQ}
                    planner.setup_optimization(goal_size=n_goals, use_standoff=use_so, axis_standoff=axis)
                    b = Ctx.last
                    fg_all[i] = b.mmin_args[0]
                    assert abs(b.costs["cost_pos"] - fg_all[i].min()) == 0.0
                    fo[i], fv[i] = b.costs["cost_obstacle"], b.costs["min_join_vel"]
                tag = f"{robot}_obj_n{n_goals}_so{int(use_so)}_{field_kind}"
                out[f"{tag}_RT"], out[f"{tag}_base"] = RT, base
                out[f"{tag}_f_goal_each"], out[f"{tag}_f_obs"], out[f"{tag}_f_vel"] = fg_all, fo, fv
                cases.append(tag)
    out[f"{robot}_Q"] = Qs
    out[f"{robot}_standoff"] = np.asarray(planner.pose_standoff)
    out[f"{robot}_dt"] = planner.dt
    b = Ctx.last
    out[f"{robot}_constraint_calls"] = np.array(repr([(c[0], c[1]) for c in b.calls]))
    out[f"{robot}_solver_options"] = np.array(repr(FakeSolver.last.options))

    # ---------------- plan_goalset / plan: seed selection and the parameter dictionaries (:145-245)
    sel = {}
    for interp in (True, False):
        n_sol = 6
        q_solutions = rng.uniform(lo, hi, size=(n_sol, R.ndof)).T.astype(np.float32)  # float32 as the driver passes them
        RTs = np.array([np.asarray(R.m.get_global_link_transform(link_ee, q_solutions[:, g].astype(np.float64))) for g in range(n_sol)])
        base = np.array([0.03, -0.02, 0.01])
        planner = GTOPlanner(R, link_ee, link_gr)
        Ctx.values = {"qc": qc, "tf_goal": np.zeros((16, n_sol)), "sdf_cost_all": fields["sparse_all"],
                      "sdf_cost_obstacle": fields["sparse_obs"], "base_position": base, f"{rn}/q": Qs[0], f"{rn}/dq": np.zeros((R.ndof, T - 1))}
        Qr, dQr, fr = planner.plan_goalset(qc, RTs, fields["sparse_all"], fields["sparse_obs"], base, q_solutions=q_solutions,
                                           use_standoff=True, axis_standoff=axis, interpolate=interp)
        s = FakeSolver.last
        tag = f"{robot}_seed_interp{int(interp)}"
        out[f"{tag}_q_solutions"], out[f"{tag}_RTs"], out[f"{tag}_base"] = q_solutions, RTs, base
        out[f"{tag}_Q0"] = Qr  # the seed incl. parameter rows, as handed to the solver
        out[f"{tag}_tf_goal"] = s.params["tf_goal"]
        out[f"{tag}_param_keys"] = np.array(sorted(s.params.keys()))
        # scores of every candidate, by the reference's compute_plan_cost
        plans = []
        for i in range(n_sol):
            data = ref.utils.interpolate_waypoints(np.stack([qc, q_solutions[:, i]]), T, R.ndof)
            data[:, par] = qc[par]
            plans.append(data.T.copy())
        cd = np.array([R.compute_plan_cost(p, fields["sparse_obs"], base) for p in plans])
        out[f"{tag}_cost_all"], out[f"{tag}_dist_all"] = cd[:, 0], cd[:, 1]
        out[f"{tag}_index"] = int(np.lexsort((cd[:, 1], cd[:, 0]))[0])
    planner = GTOPlanner(R, link_ee, link_gr)
    q_sol = rng.uniform(lo, hi)
    RT1 = np.asarray(R.m.get_global_link_transform(link_ee, q_sol))
    Ctx.values.update({"tf_goal": np.zeros((16, 1))})
    Qr, _, _ = planner.plan(qc, RT1, fields["sparse_obs"], np.zeros(3), q_solution=q_sol, use_standoff=True, axis_standoff=axis)
    out[f"{robot}_plan_param_keys"] = np.array(sorted(FakeSolver.last.params.keys()))  # no sdf_cost_all (:165-173)
    out[f"{robot}_plan_q_solution"], out[f"{robot}_plan_Q0"] = q_sol, Qr

    # ---------------- IK objective (gto/ik_solver.py:30-76)
    nq = 12
    q_ik = rng.uniform(lo, hi, size=(nq, R.ndof))
    q_ik[:, par] = qc[par]
    RT_ik = np.array([np.asarray(R.m.get_global_link_transform(link_ee, rng.uniform(lo, hi))) for _ in range(nq)])
    base_ik = rng.uniform(-0.05, 0.05, size=(nq, 3))
    ik_pos, ik_obs = np.zeros(nq), np.zeros(nq)
    for i in range(nq):
        ik = IKSolver(R, link_ee, link_gr, collision_avoidance=True)
        Ctx.values = {"tf_goal": RT_ik[i], "sdf_cost_obstacle": fields["dense_obs"], "base_position": base_ik[i], f"{rn}/q": q_ik[i].reshape(-1, 1)}
        ik.setup_optimization()
        ik_pos[i], ik_obs[i] = Ctx.last.costs["cost_pos"], Ctx.last.costs["cost_obstacle"]
    out[f"{robot}_ik_q"], out[f"{robot}_ik_RT"], out[f"{robot}_ik_base"] = q_ik, RT_ik, base_ik
    out[f"{robot}_ik_cost_pos"], out[f"{robot}_ik_cost_obstacle"] = ik_pos, ik_obs
    out[f"{robot}_ik_solver_options"] = np.array(repr(FakeSolver.last.options))

    # ---------------- base-placement objective (gto/base_planner.py:35-93)
    for n_goals in (1, 3):
        nb = 6
        ys = np.c_[rng.uniform(-0.4, 0.4, nb), rng.uniform(-0.4, 0.4, nb), rng.uniform(-0.7, 0.7, nb)]
        Qb = np.transpose(rng.uniform(lo, hi, size=(nb, n_goals, R.ndof)), (0, 2, 1)).copy()
        Qb[:, par, :] = qc[par][None, :, None]
        RTb = np.array([[np.asarray(R.m.get_global_link_transform(link_ee, rng.uniform(lo, hi))) for _ in range(n_goals)] for _ in range(nb)])
        eff, pos = np.zeros(nb), np.zeros(nb)
        for i in range(nb):
            bp = BasePlanner(R, link_ee, link_gr)
            y = np.zeros((3, n_goals))
            y[:, 0] = ys[i]
            Ctx.values = {"tf_goal": np.stack([RTb[i, g].flatten() for g in range(n_goals)], axis=1),
                          f"{rn}/q": Qb[i], f"{bp.task_name}/y": y}
            bp.setup_optimization(goal_size=n_goals, base_effort_weight=0.01)
            eff[i], pos[i] = Ctx.last.costs["cost_effort"], Ctx.last.costs["cost_pos"]
        out[f"{robot}_base_n{n_goals}_y"], out[f"{robot}_base_n{n_goals}_Q"], out[f"{robot}_base_n{n_goals}_RT"] = ys, Qb, RTb
        out[f"{robot}_base_n{n_goals}_cost_effort"], out[f"{robot}_base_n{n_goals}_cost_pos"] = eff, pos
        out[f"{robot}_base_calls"] = np.array(repr([(c[0], c[1], c[2]) for c in Ctx.last.calls if c[0] == "bound"]))
    return cases


def main():
    ref = stubs.install()
    rng = np.random.default_rng(20240930)
    out = {}
    cases = []
    with contextlib.redirect_stdout(io.StringIO()):
        for robot in ("panda", "fetch"):
            cases += run_robot(ref, robot, rng, out)
    out["cases"] = np.array(cases)
    np.savez_compressed(f"{HERE}/objective.npz", **out)
    print("objective.npz", os.path.getsize(f"{HERE}/objective.npz"), "bytes,", len(cases), "objective cases")


if __name__ == "__main__":
    main()
