"""CPU: the bookkeeping of the device-resident per-object perception (grasptrajopt_amd/depth_scene.py, gto_models.py) with a
recording stand-in for the solver handles: which images go into the ONE build of the resident scene when the calls are
made the way the reference's driver makes them (examples/pybullet_gto_planning.py:176-190, 242-272, 291), which half of
that scene every consumer reads, and when a borrowed scene is shared again.  No GPU: nothing here computes a field."""
import numpy as np
import pytest

import grasptrajopt_amd as g
from grasptrajopt_amd import depth_scene as ds
from grasptrajopt_amd import optas_facade as optas


class FakeHandle:
    """Records what the Python layer asks of a solver handle; scene generations as _capi.SolverHandle keeps them."""

    def __init__(self, name):
        self.name, self.calls, self._gen, self.scenes, self.T = name, [], {}, {}, 50

    def _bump(self, sid):
        self._gen[sid] = self._gen.get(sid, 0) + 1

    def scene_generation(self, sid):
        return self._gen.get(sid, 0)

    def scene_from_depth(self, sid, depth, K, cam, target_mask=None, threshold=1.5, grid_res=0.05, margin=0.4, epsilon=0.02, w_inside=1.0,
                         depth_obstacle=None):
        self.calls.append(("build", np.array(depth), None if target_mask is None else np.array(target_mask),
                           None if depth_obstacle is None else np.array(depth_obstacle), float(threshold)))
        self._bump(sid)
        self.scenes[sid] = ((4, 5, 6), np.zeros(3), grid_res)
        return (4, 5, 6), np.zeros(3), np.array([[0.0, 1.0], [0.0, 1.0], [0.0, 1.0]])

    def share_scene(self, sid, src, src_sid=None, all_from=0, obs_from=1):
        self.calls.append(("share", src.name, src_sid, all_from, obs_from, src.scene_generation(src_sid)))
        self._bump(sid)

    def set_scene(self, sid, c_all, c_obs, shape, origin, res, values_only=False):
        self.calls.append(("set_scene", sid))
        self._bump(sid)

    def set_opts(self, **kw):
        pass

    def scene_fields(self, sid):
        n = 4 * 5 * 6
        return np.full(n, 1.0, np.float32), np.full(n, 2.0, np.float32)

    def solve_ik_batch(self, sid, q0, goals, base, max_iter):
        B = goals.shape[0]
        return np.zeros((B, 9)), np.zeros(B), np.zeros(B, np.int32), np.zeros(B, np.int32)

    def eval_fk(self, q):
        return np.tile(np.eye(4), (q.shape[0], 12, 1, 1))

    def eval_points(self, sid, q, base, use_obs=False, want=None, want_field=True):
        return None, None, np.zeros((q.shape[0], 3)), None

    def plan_cost(self, sid, plans, base):
        self.calls.append(("plan_cost", sid))
        return np.zeros(plans.shape[0]), np.zeros(plans.shape[0])

    def close(self):
        pass


@pytest.fixture()
def rig(monkeypatch):
    robot = g.GTORobotModel(desc=g.load_builtin("panda"), time_derivs=[0, 1], param_joints=["panda_finger_joint1", "panda_finger_joint2"])
    handles = {}

    def solver_handle(link_ee, link_gripper, opts=None, role="planner"):
        return handles.setdefault(role, FakeHandle(role))
    monkeypatch.setattr(robot, "solver_handle", solver_handle)
    H, W = 12, 16
    depth = np.linspace(0.8, 1.2, H * W, dtype=np.float32).reshape(H, W)
    mask = np.zeros((H, W), np.uint8)
    mask[3:6, 4:9] = 1
    K = np.array([[20.0, 0, 8.0], [0, 20.0, 6.0], [0, 0, 1.0]])
    cam = np.eye(4)
    return robot, handles, depth, mask, K, cam


def _driver_fields(robot, depth, mask, K, cam, thr=1.5):
    """The driver's perception calls (examples/pybullet_gto_planning.py:176-190)."""
    depth_pc = g.DepthPointCloud(depth, K, cam, target_mask=None, threshold=thr)
    robot.setup_points_field(depth_pc.points)
    world_points = robot.workspace_points
    sdf_cost_all = depth_pc.get_sdf_cost(world_points)
    depth_obstacle = depth.copy()
    depth_obstacle[mask.astype(bool)] = thr
    depth_pc_obstacle = g.DepthPointCloud(depth_obstacle, K, cam, mask, threshold=thr)
    sdf_cost_obstacle = depth_pc_obstacle.get_sdf_cost(world_points)
    return sdf_cost_all, sdf_cost_obstacle, depth_obstacle


def test_driver_pattern_builds_one_scene_from_both_images(rig):
    robot, handles, depth, mask, K, cam = rig
    la, lo, depth_obstacle = _driver_fields(robot, depth, mask, K, cam)
    assert isinstance(la, ds.LazyCostField) and isinstance(lo, ds.LazyCostField)
    ik = g.IKSolver(robot, "panda_hand", "panda_hand", collision_avoidance=True)
    ik.solve_ik_batch(np.zeros(9), np.tile(np.eye(4), (3, 1, 1)), lo, [0, 0, 0])     # the first consumer: builds the scene
    util = handles["util"]
    builds = [c for c in util.calls if c[0] == "build"]
    assert len(builds) == 1
    _, d_all, m, d_obs, thr = builds[0]
    np.testing.assert_array_equal(d_all, depth)            # field 0 (and the grid): the image of ALL pixels, target included
    np.testing.assert_array_equal(m, mask)
    np.testing.assert_array_equal(d_obs, depth_obstacle)   # field 1: the obstacle image (its visibility test reads it)
    assert handles["ik"].calls[-1][:5] == ("share", "util", ds.DEPTH_SCENE, 1, 1)   # the IK reads the obstacle field as both halves
    # the planner: both fields of the SAME build, each as the half it is
    planner = g.GTOPlanner(robot, "panda_hand", "panda_hand")
    planner.setup_optimization(goal_size=1)
    planner.solver.reset_parameters({"qc": np.zeros(9), "tf_goal": np.eye(4).reshape(16, 1), "sdf_cost_all": la, "sdf_cost_obstacle": lo,
                                     "base_position": np.zeros(3), "panda/q/p": np.zeros((2, 50))})
    planner.solver.ensure_scene()
    assert len([c for c in util.calls if c[0] == "build"]) == 1          # still one build
    assert handles["planner"].calls[-1][:5] == ("share", "util", ds.DEPTH_SCENE, 0, 1)
    # the field of all pixels handed to an entry point that reads "the obstacle field": it is that field that is read
    ik.solve_ik_batch(np.zeros(9), np.tile(np.eye(4), (1, 1, 1)), la, [0, 0, 0])
    assert handles["ik"].calls[-1][:5] == ("share", "util", ds.DEPTH_SCENE, 0, 0)
    robot.compute_plan_cost(np.zeros((9, 50)), la, [0, 0, 0])
    assert ("share", "util", ds.DEPTH_SCENE, 0, 0) == util.calls[-2][:5] and util.calls[-1] == ("plan_cost", robot.SCRATCH_SCENE)
    robot.compute_plan_cost(np.zeros((9, 50)), lo, [0, 0, 0])
    assert util.calls[-1] == ("plan_cost", ds.DEPTH_SCENE)               # the obstacle half of the handle's own scene: read in place
    assert len([c for c in util.calls if c[0] == "build"]) == 1
    # as arrays: each lazy field is the half it stands for
    assert float(np.asarray(la)[0]) == 1.0 and float(np.asarray(lo)[0]) == 2.0


def test_borrowers_share_again_after_a_rebuild(rig):
    robot, handles, depth, mask, K, cam = rig
    la, lo, _ = _driver_fields(robot, depth, mask, K, cam)
    planner = g.GTOPlanner(robot, "panda_hand", "panda_hand")
    lo.resident()                       # (plan_goalset's first step: the obstacle field's build holds both fields and the grid)
    planner.setup_optimization(goal_size=1)
    pd = {"qc": np.zeros(9), "tf_goal": np.eye(4).reshape(16, 1), "base_position": np.zeros(3), "panda/q/p": np.zeros((2, 50))}
    planner.solver.reset_parameters(dict(pd, sdf_cost_all=la, sdf_cost_obstacle=lo))
    planner.solver.ensure_scene()
    util, ph = handles["util"], handles["planner"]
    n_share = len([c for c in ph.calls if c[0] == "share"])
    planner.solver.ensure_scene()                                        # nothing changed: nothing is shared again
    assert len([c for c in ph.calls if c[0] == "share"]) == n_share
    # another object's image: the resident scene is rebuilt in place
    depth2 = depth + np.float32(0.05)
    la2, lo2, _ = _driver_fields(robot, depth2, mask, K, cam)
    lo2.resident()
    assert len([c for c in util.calls if c[0] == "build"]) == 2
    # the first object's solver borrowed the build that is gone: its next use builds its own scene again and shares that
    planner.solver.ensure_scene()
    builds = [c for c in util.calls if c[0] == "build"]
    assert len(builds) == 3
    np.testing.assert_array_equal(builds[2][1], depth)
    last = ph.calls[-1]
    assert last[:5] == ("share", "util", ds.DEPTH_SCENE, 0, 1) and last[5] == util.scene_generation(ds.DEPTH_SCENE)
    # an image edited in place after the build is another image (the cache compares values, it keeps copies)
    depth[0, 0] += np.float32(0.25)
    assert la.resident().gen == util.scene_generation(ds.DEPTH_SCENE)
    assert len([c for c in util.calls if c[0] == "build"]) == 4
    n_build = len([c for c in util.calls if c[0] == "build"])
    la.resident(), la.resident()                                          # unchanged inputs: the stamp answers, nothing is built or compared
    assert len([c for c in util.calls if c[0] == "build"]) == n_build


def test_in_place_ufunc_output_makes_a_field_the_callers_own(rig):
    """np.clip(field, 0, 1, out=field), field *= 2: written in place through __array_ufunc__ -- the array the caller sees is no
    longer the resident scene's field, so the field stops naming it (only __setitem__ marked the edit)."""
    robot, handles, depth, mask, K, cam = rig
    la, lo, _ = _driver_fields(robot, depth, mask, K, cam)
    assert lo.resident() is not None and la.resident() is not None
    np.clip(lo, 0.0, 1.0, out=lo)
    assert lo.resident() is None and float(np.asarray(lo).max()) <= 1.0
    seen_elsewhere = la                     # (a solver's parameter dictionary holds the same object)
    la *= 2                                 # the name now holds the plain array numpy returned ...
    assert isinstance(la, np.ndarray) and ds.resident_of(la) is None
    assert seen_elsewhere.resident() is None and float(np.asarray(seen_elsewhere)[0]) == 2.0  # ... and the object is marked edited


def test_fields_that_cannot_share_one_build_go_through_the_host(rig):
    robot, handles, depth, mask, K, cam = rig
    la, lo, _ = _driver_fields(robot, depth, mask, K, cam)
    cam2 = np.eye(4)
    cam2[0, 3] = 0.1
    other = g.DepthPointCloud(depth, K, cam2, mask, threshold=1.5).get_sdf_cost(robot.workspace_points)   # another camera
    assert other.resident() is None
    other_thr = g.DepthPointCloud(depth, K, cam, mask, threshold=1.2).get_sdf_cost(robot.workspace_points)  # another cut-off
    assert other_thr.resident() is None
    lo[3] = 7.0                                                           # edited: an array of the caller's from now on
    assert lo.resident() is None and float(np.asarray(lo)[3]) == 7.0
    planner = g.GTOPlanner(robot, "panda_hand", "panda_hand")
    robot.field_size = 4 * 5 * 6
    planner.setup_optimization(goal_size=1)
    planner.solver.reset_parameters({"qc": np.zeros(9), "tf_goal": np.eye(4).reshape(16, 1), "sdf_cost_all": la, "sdf_cost_obstacle": lo,
                                     "base_position": np.zeros(3), "panda/q/p": np.zeros((2, 50))})
    planner.solver.ensure_scene()
    assert handles["planner"].calls[-1] == ("set_scene", 0)


def test_a_grid_given_as_numbers_forgets_the_pending_cloud(rig):
    robot, handles, depth, mask, K, cam = rig
    dpc = g.DepthPointCloud(depth, K, cam, target_mask=None, threshold=1.5)
    robot.setup_points_field(dpc.points)
    assert robot.__dict__.get("_pending_depth") is dpc
    robot.setup_workspace_field(0.5, 0.3)
    assert robot.__dict__.get("_pending_depth") is None
    wp = robot.workspace_points
    assert isinstance(wp, np.ndarray) and wp.shape[0] == robot.field_size
    # a masked cloud never stays pending: the resident build sizes the grid from the cloud of all pixels
    robot.setup_points_field(np.array([[0.0, 0.0, 0.0], [0.3, 0.2, 0.1]]))
    assert robot.__dict__.get("_pending_depth") is None and isinstance(robot.workspace_points, np.ndarray)


class _Lazy(ds._LazyArray):
    def __init__(self, a):
        self.a, self.n = a, 0

    def _materialize(self):
        self.n += 1
        return self.a.copy()


def test_lazy_stand_ins_behave_as_arrays():
    a = np.array([1.0, -2.0, 3.0], np.float32)
    z = _Lazy(a)
    np.testing.assert_array_equal(z * 2, a * 2)
    np.testing.assert_array_equal(2 * z, a * 2)
    np.testing.assert_array_equal(-z, -a)
    np.testing.assert_array_equal(z - a, np.zeros(3, np.float32))
    np.testing.assert_array_equal(a - z, np.zeros(3, np.float32))
    np.testing.assert_array_equal(z < 0, a < 0)
    np.testing.assert_array_equal(z == a, np.ones(3, bool))
    np.testing.assert_array_equal(np.add(z, 1), a + 1)
    np.testing.assert_array_equal(np.abs(z), np.abs(a))
    assert float(np.sum(z)) == 2.0 and z.shape == (3,) and z.dtype == np.float32 and len(z) == 3
    assert [float(x) for x in z] == [1.0, -2.0, 3.0]
    z[z < 0] = 0.0
    np.testing.assert_array_equal(np.asarray(z), [1.0, 0.0, 3.0])
    assert isinstance(z + z, np.ndarray) and z.n == 1   # materialised once
    out = np.empty(3, np.float32)
    np.multiply(z, 2, out=out)
    np.testing.assert_array_equal(out, [2.0, 0.0, 6.0])
