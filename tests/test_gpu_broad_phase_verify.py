"""The step kernel's broad phase checked by the obstacle kernel itself: a build of the library with -DGTO_DEBUG_LONGEST_WG run with
GTO_DEBUG_CUT=10 lists EVERY waypoint group for the obstacle launch, the ones the step kernel settled marked as such, and counts the
marked groups that nevertheless receive a contribution from a non-zero voxel record.  That count has to be zero: the step kernel's
test (single-precision kinematics, spheres over runs of chunks, widened radii: gto_kernels.h, prebroad_tail) may keep more than the
obstacle kernel's exact per-chunk test, never less than the points need.  Results are unchanged by the listing (checked here too)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["GTO_ROOT"])
from grasptrajopt_amd import _capi, synthetic as syn
from grasptrajopt_amd.robot_desc import load_builtin
robot, B, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
name = "panda" if robot.startswith("panda") else "fetch"
cfg = json.load(open(os.path.join(os.environ["GTO_ROOT"], "grasptrajopt_amd", "data", name + "_cfg.json")))
d = load_builtin(robot)
opts = _capi.default_opts(); opts.max_iter = 60
h = _capi.SolverHandle(d, cfg["link_ee"], cfg["link_gripper"], opts, device=0)
T = opts.T
qc = np.tile(np.array(cfg["default_pose"], dtype=np.float64), (B, 1))
if name == "panda":
    sc = syn.make_scene(0, n=128, res=2.24 / 128, origin=(-0.4, -1.12, -0.4), table_z=-0.03)
    h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    RT, qg = syn.make_goals(d, h.eval_fk, cfg["link_ee"], B, seed=3, zlim=(0.08, 0.7))
    Q0 = np.stack([syn.make_seed(qc[b], qg[b], T, d.param_index) for b in range(B)])
else:  # the shelf of BASELINE configs[2], seeds of interpolate=False
    sc = syn.make_scene(11, n=128, res=2.24 / 128, origin=(-0.3, -1.12, 0.0), table_z=0.75, shelf=True)
    h.set_scene(0, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    RT, qg = syn.make_goals(d, h.eval_fk, cfg["link_ee"], B, seed=5, xlim=(0.45, 0.85), ylim=(-0.45, 0.45), zlim=(0.82, 1.08))
    qg[:, d.param_index] = qc[:, d.param_index]
    Q0 = np.repeat(qc[:, :, None], T, axis=2)
    Q0[:, :, T - 10:] = qg[:, :, None]
S = np.tile(syn.standoff_pose(-0.1, cfg["axis_standoff"]).reshape(1, 16), (B, 1))
Q, dQ, f, it, st = h.solve_batch(0, qc, RT.reshape(B, 1, 16), 1, S, np.zeros((B, 3)), Q0)
np.savez(out, Q=Q, f=f, it=it, st=st)
'''


@pytest.fixture(scope="module")
def debug_library(tmp_path_factory):
    lib = str(tmp_path_factory.mktemp("dbg") / "libgto_hip_dbg.so")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    src = [os.path.join(g.CSRC, s) for s in g.HIP_SOURCES]
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + g.HIPCC_FLAGS + ["-DGTO_DEBUG_LONGEST_WG"] + src + ["-o", lib],
                          cwd=g.CSRC, stderr=subprocess.DEVNULL)
    return lib


def _run(robot, B, out, env_extra):
    env = dict(os.environ, GTO_ROOT=ROOT, **env_extra)
    r = subprocess.run([sys.executable, "-c", WORKER, robot, str(B), out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stderr


@pytest.mark.parametrize("robot,B", [("panda_5k", 512), ("fetch", 320)])
def test_groups_settled_by_the_step_kernel_get_no_contribution(debug_library, tmp_path, robot, B):
    a, b = str(tmp_path / "verify.npz"), str(tmp_path / "plain.npz")
    err = _run(robot, B, a, {"GTO_HIP_LIB": debug_library, "GTO_DEBUG_TIMING": "1", "GTO_DEBUG_CUT": "10"})
    lines = [l for l in err.splitlines() if "groups settled" in l]
    assert lines, err[-2000:]
    settled = contributions = 0
    for l in lines:
        m = re.search(r"(\d+) groups settled, (\d+) of them with a surviving chunk, (\d+) with a CONTRIBUTION", l)
        assert m, l
        settled += int(m.group(1))
        contributions += int(m.group(3))
    assert settled > 1000, lines  # the broad phase ran and settled groups (B instances in flight: the rounds that fill the GPU)
    assert contributions == 0, lines
    _run(robot, B, b, {})  # the shipped library, nothing listed that does not have to be
    va, vb = np.load(a), np.load(b)
    for k in ("Q", "f", "it", "st"):
        np.testing.assert_array_equal(va[k], vb[k])
