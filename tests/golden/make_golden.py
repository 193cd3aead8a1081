#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by EXECUTING THE REFERENCE'S OWN PYTHON
numerics from /root/reference.  Build-container only; the fixtures
(.npz, data only) are what travels.  Re-run:  python tests/golden/make_golden.py

What executes from the reference:
  optas/spatialmath.py + optas/models.py   RobotModel FK, limits, joint bookkeeping
  gto/gto_models.py                        setup_points_field, points_to_offsets_numpy,
                                           setup_fk_functions' visual-origin composition (function
                                           bodies extracted with ast and exec'd: the module itself
                                           imports trimesh/pyrender/turtle and cannot be imported)
  gto/sdf_callback.py                      SDFCallback / JacFun / HesFun .eval
  gto/utils.py                             interpolate_waypoints (scipy CubicSpline)
  mesh_to_sdf/depth_point_cloud.py         DepthPointCloud.get_sdf / get_sdf_cost (sklearn KDTree)
  gto/gto_models.py                        setup_occupancy_grid, points_to_offsets_occupancy_numpy (mobile base)
Third-party modules that are absent (casadi, urdf_parser_py, ...) are replaced by the numpy
stand-ins in _reference_stubs.py; no reference source is copied into the fixtures.
"""
import ast
import glob
import json
import os
import sys
import textwrap
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference_stubs as stubs  # noqa: E402

REF = stubs.REF


def extract_functions(path, names, extra_ns=None):
    """Compile selected function definitions out of a reference file without importing it."""
    src = open(path).read()
    tree = ast.parse(src)
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = textwrap.dedent(ast.get_source_segment(src, node))
            ns = {"np": np}
            ns.update(extra_ns or {})
            exec(compile(code, path, "exec"), ns)
            out[node.name] = ns[node.name]
    return out


def golden_fk(ref, robot, rng):
    cfg = yaml.safe_load(open(f"{REF}/data/configs/{robot}.yaml"))["robot_cfg"]
    m = ref.models.RobotModel(urdf_filename=f"{REF}/{cfg['urdf_robot_path']}", time_derivs=[0, 1],
                              param_joints=cfg["param_joints"])
    lo = np.asarray(m.lower_actuated_joint_limits).ravel()
    hi = np.asarray(m.upper_actuated_joint_limits).ravel()
    lo_c, hi_c = np.maximum(lo, -3.2), np.minimum(hi, 3.2)  # continuous joints have +-1e9
    nq = 32
    q = rng.uniform(lo_c, hi_c, size=(nq, m.ndof))
    q[0] = np.array(cfg["default_pose"])
    q[1] = 0.0
    link_names = m.link_names
    frames = np.zeros((nq, len(link_names), 4, 4))
    for i in range(nq):
        for j, ln in enumerate(link_names):
            frames[i, j] = np.asarray(m.get_global_link_transform(ln, q[i]))
    # visual_tf = link_tf @ rt2tr(rpy2r(rpy), xyz)   (gto/gto_models.py:92-100, executed via the
    # reference's spatialmath functions on the reference's get_link_visual_origin output)
    urdf = m.get_urdf()
    vis_names = [l.name for l in urdf.links if l.visual is not None
                 and l.name in cfg["collision_link_names"]]
    visual = np.zeros((nq, len(vis_names), 4, 4))
    for j, ln in enumerate(vis_names):
        xyz, rpy = m.get_link_visual_origin(urdf.link_map[ln])
        V = np.asarray(ref.spatialmath.rt2tr(ref.spatialmath.rpy2r(rpy), xyz))
        for i in range(nq):
            visual[i, j] = frames[i, link_names.index(ln)] @ V
    gripper_tf = np.asarray(m.get_link_transform(cfg["link_gripper"], q[0], cfg["link_ee"]))
    return dict(q=q, link_names=np.array(link_names), frames=frames, visual_names=np.array(vis_names),
                visual=visual, gripper_tf=gripper_tf, lower=lo, upper=hi,
                opt_index=np.array(m.optimized_joint_indexes), param_index=np.array(m.parameter_joint_indexes),
                actuated=np.array(m.actuated_joint_names), ndof=m.ndof,
                lower_opt=np.asarray(m.lower_optimized_joint_limits).ravel(),
                upper_opt=np.asarray(m.upper_optimized_joint_limits).ravel())


def golden_sdf(ref, rng):
    shape = (16, 20, 12)
    origin = np.array([-0.31, -0.52, 0.07])
    res = 0.05
    data32 = rng.random(shape).astype(np.float32).reshape(-1)
    data = data32.astype(np.float64)
    lo = origin - 3 * res
    hi = origin + (np.array(shape) + 3) * res
    pts = rng.uniform(lo, hi, size=(256, 3))
    # exact grid nodes and points a hair either side of voxel faces
    ijk = rng.integers(0, np.array(shape), size=(32, 3))
    pts[:32] = origin + ijk * res
    pts[32:48] = origin + rng.integers(0, np.array(shape), size=(16, 3)) * res + 1e-12
    pts[48:64] = origin + rng.integers(0, np.array(shape), size=(16, 3)) * res - 1e-12
    f = ref.sdf_callback.SDFCallback("f", data, origin, res, shape)
    jf = f.get_jacobian("jac_f", None, None, {})
    hf = jf.get_jacobian("hes_f", None, None, {})
    val = np.zeros(len(pts))
    jac = np.zeros((len(pts), 3))
    hes = np.zeros((len(pts), 3, 3))
    for i, p in enumerate(pts):
        val[i] = np.asarray(f.eval([p])[0]).item()
        jac[i] = np.asarray(jf.eval([p, None])[0]).reshape(3)
        hes[i] = np.asarray(hf.eval([p, None, None])[0])
    return dict(shape=np.array(shape), origin=origin, res=res, data=data32, points=pts, value=val, jac=jac, hess=hes)


def golden_grid(rng):
    fns = extract_functions(f"{REF}/gto/gto_models.py", {"setup_points_field", "points_to_offsets_numpy"})
    out = {}
    cloud = rng.uniform([0.2, -0.6, -0.05], [0.9, 0.55, 0.8], size=(500, 3))
    obj = types.SimpleNamespace(field_margin=0.4, grid_resolution=0.05)
    import builtins
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        fns["setup_points_field"](obj, cloud)
    out.update(cloud=cloud, origin=obj.origin, field_shape=np.array(obj.field_shape),
               field_size=obj.field_size, workspace_points=obj.workspace_points.astype(np.float64))
    lo = obj.origin.ravel() - 0.2
    hi = obj.origin.ravel() + (np.array(obj.field_shape) + 4) * 0.05
    q = rng.uniform(lo, hi, size=(2000, 3))
    q[:200] = obj.workspace_points[rng.integers(0, obj.field_size, 200)]  # exact grid nodes
    out["query"] = q
    out["offsets"] = fns["points_to_offsets_numpy"](obj, q.copy())
    return out


def golden_occupancy(rng):
    """x-y occupancy grid of the mobile pipeline (gto/gto_models.py:218-270; sklearn KDTree as there)."""
    from sklearn.neighbors import KDTree
    import io
    import contextlib
    fns = extract_functions(f"{REF}/gto/gto_models.py", {"setup_occupancy_grid", "points_to_offsets_occupancy_numpy"},
                            {"KDTree": KDTree})
    # a table-like slab of points in front of the robot plus floor points (z <= 0.01 are ignored by the reference)
    n = 6000
    cloud = np.concatenate([
        np.c_[rng.uniform(0.6, 1.4, n), rng.uniform(-0.7, 0.5, n), rng.uniform(0.02, 0.8, n)],
        np.c_[rng.uniform(0.0, 2.0, 500), rng.uniform(-1.0, 1.0, 500), rng.uniform(-0.01, 0.01, 500)],
        np.c_[rng.uniform(1.7, 1.75, 40), rng.uniform(0.9, 0.95, 40), rng.uniform(0.3, 0.4, 40)]])
    obj = types.SimpleNamespace(field_margin=0.4, grid_resolution=0.05)
    with contextlib.redirect_stdout(io.StringIO()):
        fns["setup_occupancy_grid"](obj, cloud)
    q = np.c_[rng.uniform(-0.8, 2.6, 3000), rng.uniform(-1.6, 1.8, 3000), rng.uniform(0, 1, 3000)]
    return dict(cloud=cloud, origin=obj.occupancy_grid_origin, shape=np.array(obj.occupancy_grid_shape),
                size=obj.occupancy_grid_size, grid=obj.occupancy_grid, query=q,
                offsets=fns["points_to_offsets_occupancy_numpy"](obj, q.copy()))


def golden_depth(ref, rng):
    H, W = 48, 64
    K = np.array([[60.0, 0, 32.0], [0, 60.0, 24.0], [0, 0, 1.0]])
    depth = np.full((H, W), 1.0, dtype=np.float32)
    depth[14:34, 20:44] = 0.7  # a box in front of a wall
    depth[0:3, :] = 0.0        # invalid pixels
    cam = np.eye(4)
    cam[:3, :3] = np.array([[0, 0, 1.0], [-1.0, 0, 0], [0, -1.0, 0]])
    cam[:3, 3] = [-0.2, 0.0, 0.5]
    mask = np.zeros((H, W), dtype=np.uint8)
    mask[20:28, 28:36] = 1
    import io
    import contextlib
    out = {}
    for tag, tm in (("all", None), ("obs", mask)):
        dpc = ref.dpc.DepthPointCloud(depth, K, cam, target_mask=tm, threshold=1.5)
        q = rng.uniform([0.0, -0.6, 0.0], [1.0, 0.6, 1.0], size=(400, 3))
        with contextlib.redirect_stdout(io.StringIO()):
            sdf = dpc.get_sdf(q)
            cost = dpc.get_sdf_cost(q, epsilon=0.02, w_inside=1)
        inside = ~dpc.is_outside(q)
        out[f"{tag}_points"] = dpc.points
        out[f"{tag}_query"] = q
        out[f"{tag}_sdf"] = sdf
        out[f"{tag}_inside"] = inside
        out[f"{tag}_cost"] = cost
    out.update(depth=depth, K=K, cam=cam, mask=mask)
    return out


def golden_interp(ref, rng, fk):
    lo, hi = np.maximum(fk["lower"], -3.2), np.minimum(fk["upper"], 3.2)
    qc = rng.uniform(lo, hi, size=(6, len(lo)))
    qg = rng.uniform(lo, hi, size=(6, len(lo)))
    seeds = np.stack([ref.utils.interpolate_waypoints(np.stack([a, b]), 50, len(lo)) for a, b in zip(qc, qg)])
    seeds7 = np.stack([ref.utils.interpolate_waypoints(np.stack([a, b]), 7, len(lo)) for a, b in zip(qc, qg)])
    return dict(qc=qc, qgoal=qg, seeds=seeds, seeds7=seeds7)


def golden_plans():
    """Structural invariants + a sample of the 853 stored GTO plans (examples/results_iros2024)."""
    out = {}
    stats = {}
    for fn in sorted(glob.glob(f"{REF}/examples/results_iros2024/GTO_*.json")):
        tag = os.path.basename(fn).split("_24-")[0].replace("GTO_scenereplica_", "")
        plans, times = [], []
        data = json.load(open(fn))
        for scene in data.values():
            for order in scene.values():
                for obj in order.values():
                    if isinstance(obj, dict) and obj.get("plan") is not None:
                        p = np.array(obj["plan"], dtype=np.float64)
                        if p.ndim == 2 and p.shape[1] == 50:
                            plans.append(p)
                            times.append(obj.get("planning_time", np.nan))
        plans = np.stack(plans)
        stats[tag] = dict(n=int(len(plans)), max_q1_q0=float(np.abs(plans[:, :, 1] - plans[:, :, 0]).max()),
                          mean_planning_time=float(np.nanmean(times)))
        out[f"{tag}_sample"] = plans[:: max(1, len(plans) // 12)][:12]
    out["stats_json"] = np.array(json.dumps(stats))
    return out


def golden_results():
    """Evaluator totals of the stored result files (examples/pybullet_evaluate_plans.py:162-181,262-290),
    counted here with plain loops, and a one-scene excerpt of one file as a wire-format sample."""
    def count(data):
        n = succ = 0
        tsum = {"checking_time": 0.0, "ik_time": 0.0, "planning_time": 0.0}
        tcnt = {"checking_time": 0, "ik_time": 0, "planning_time": 0}
        per_obj = {}
        for scene in data.values():
            for order in scene.values():
                for name, obj in order.items():
                    if not (isinstance(obj, dict) and "reward" in obj):
                        continue
                    n += 1
                    succ += obj["reward"]
                    c = per_obj.setdefault(name, [0, 0])
                    c[0] += 1
                    c[1] += obj["reward"]
                    for k in tsum:
                        if obj.get(k) is not None:
                            tsum[k] += obj[k]
                            tcnt[k] += 1
        return dict(trials=n, success=succ, per_object=per_obj,
                    mean_time={k: (tsum[k] / tcnt[k] if tcnt[k] else None) for k in tsum})

    totals = {}
    for fn in sorted(glob.glob(f"{REF}/examples/results_iros2024/GTO_*.json")):
        totals[os.path.basename(fn)] = count(json.load(open(fn)))
    fn = sorted(glob.glob(f"{REF}/examples/results_iros2024/GTO_scenereplica_mobile_fetch_tabletop_*.json"))[0]
    data = json.load(open(fn))
    first = next(iter(data))
    totals["excerpt"] = count({first: data[first]})
    return totals, {first: data[first]}, os.path.basename(fn)


def main():
    if "--only-results" in sys.argv:
        totals, excerpt, name = golden_results()
        json.dump(dict(totals=totals, excerpt=excerpt, excerpt_of=name), open(f"{HERE}/results.json", "w"))
        return
    if "--only-occupancy" in sys.argv:  # added later; its own generator so the other fixtures stay byte-identical
        np.savez_compressed(f"{HERE}/occupancy.npz", **golden_occupancy(np.random.default_rng(20240207)))
        return
    ref = stubs.install()
    rng = np.random.default_rng(20240206)
    fk_panda = golden_fk(ref, "panda", rng)
    fk_fetch = golden_fk(ref, "fetch", rng)
    np.savez_compressed(f"{HERE}/fk_panda.npz", **fk_panda)
    np.savez_compressed(f"{HERE}/fk_fetch.npz", **fk_fetch)
    np.savez_compressed(f"{HERE}/sdf_callback.npz", **golden_sdf(ref, rng))
    np.savez_compressed(f"{HERE}/grid.npz", **golden_grid(rng))
    np.savez_compressed(f"{HERE}/depth_cost.npz", **golden_depth(ref, rng))
    np.savez_compressed(f"{HERE}/interp.npz", **golden_interp(ref, rng, fk_panda))
    np.savez_compressed(f"{HERE}/stored_plans.npz", **golden_plans())
    # known answers held as literals in the reference (gto/gto_planner.py:277-285 demo goal poses)
    known = dict(
        rt_fetch=np.array([[-0.05241979, -0.45344928, -0.88973933, 0.41363978],
                           [-0.27383122, -0.8502871, 0.44947574, 0.12551154],
                           [-0.96034825, 0.26719978, -0.07959669, 0.97476065], [0, 0, 0, 1.0]]),
        rt_panda=np.array([[-0.61162336, 0.79089652, 0.01998741, 0.46388378],
                           [0.7883297, 0.6071185, 0.09971584, -0.15167381],
                           [0.06673018, 0.07674521, -0.99481508, 0.22877409], [0, 0, 0, 1.0]]))
    np.savez_compressed(f"{HERE}/known_answers.npz", **known)
    np.savez_compressed(f"{HERE}/occupancy.npz", **golden_occupancy(np.random.default_rng(20240207)))
    totals, excerpt, name = golden_results()
    json.dump(dict(totals=totals, excerpt=excerpt, excerpt_of=name), open(f"{HERE}/results.json", "w"))
    for f in sorted(glob.glob(f"{HERE}/*.npz")):
        print(os.path.basename(f), os.path.getsize(f))


if __name__ == "__main__":
    main()
