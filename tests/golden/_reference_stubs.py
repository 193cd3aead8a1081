"""Numpy-backed stand-ins that let the *reference's own Python* numerics execute in this
container.  DEV-TIME ONLY: used by make_golden.py to produce the
committed fixtures under tests/golden/.  Nothing in tests/, bench.py or the package imports
this module at run time, and /root/reference never travels to the GPU box.

What is stubbed (absent third-party modules, not reference code):
  casadi           -> DM as an ndarray subclass + the dozen free functions spatialmath/models use
  urdf_parser_py   -> ElementTree reader exposing the attributes optas.models touches
  xacro, pyrender, transforms3d, _init_paths -> empty modules
The arithmetic that runs is the reference's (optas/spatialmath.py, optas/models.py,
gto/sdf_callback.py, gto/utils.py, mesh_to_sdf/depth_point_cloud.py), loaded by file path.
"""
import importlib.util
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

REF = "/root/reference"


# ----------------------------------------------------------------------------- casadi
class DM(np.ndarray):
    def __new__(cls, data=None, *rest):
        if data is None:
            a = np.zeros((0, 1))
        elif rest:
            a = np.zeros((int(data), int(rest[0])))
        else:
            a = np.array(data, dtype=np.float64)
            if a.ndim == 0:
                a = a.reshape(1, 1)
            elif a.ndim == 1:
                a = a.reshape(-1, 1)
        return a.view(cls)

    @staticmethod
    def eye(n):
        return np.eye(n).view(DM)

    @staticmethod
    def zeros(n, m=1):
        return np.zeros((n, m)).view(DM)

    @staticmethod
    def ones(n, m=1):
        return np.ones((n, m)).view(DM)

    def toarray(self):
        return np.asarray(self)


class SX(DM):
    pass


class MX(DM):
    pass


def _as2d(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 0:
        return a.reshape(1, 1)
    if a.ndim == 1:
        # casadi treats 1-D input as a column; a length-1 vector is a scalar
        return a.reshape(-1, 1)
    return a


def horzcat(*args):
    if len(args) == 1 and isinstance(args[0], (list, tuple)):
        return _as2d(args[0]).view(DM)
    return np.hstack([_as2d(a) for a in args]).view(DM)


def vertcat(*args):
    return np.vstack([_as2d(a) for a in args]).view(DM)


def vertsplit(a):
    a = _as2d(a)
    return [a[i : i + 1, :].view(DM) for i in range(a.shape[0])]


def horzsplit(a):
    a = _as2d(a)
    return [a[:, i : i + 1].view(DM) for i in range(a.shape[1])]


def vec(a):
    return _as2d(a).reshape(-1, 1, order="F").view(DM)


def norm_fro(a):
    return float(np.sqrt(np.sum(np.asarray(a) ** 2)))


class _Callback:
    def __init__(self):
        pass

    def construct(self, name, opts=None):
        self.init()

    def init(self):
        pass


class _Sparsity:
    @staticmethod
    def dense(n, m=1):
        return (n, m)


def _install_casadi():
    cs = types.ModuleType("casadi")
    cs.DM, cs.SX, cs.MX = DM, SX, MX
    cs.horzcat, cs.vertcat, cs.vertsplit, cs.horzsplit = horzcat, vertcat, vertsplit, horzsplit
    cs.vec, cs.norm_fro = vec, norm_fro
    cs.sin, cs.cos, cs.sqrt = np.sin, np.cos, np.sqrt
    cs.np = np
    cs.Function = object
    cs.Callback = _Callback
    cs.Sparsity = _Sparsity
    cs.casadi = cs
    cs.__all__ = ["DM", "SX", "MX", "Callback", "Sparsity", "horzcat", "vertcat", "vec"]
    sys.modules["casadi"] = cs
    return cs


# ----------------------------------------------------------------------------- urdf_parser_py
class _NS(types.SimpleNamespace):
    pass


def _floats(s):
    return [float(x) for x in s.split()]


def _origin(el):
    o = el.find("origin") if el is not None else None
    if o is None:
        return None
    return _NS(xyz=_floats(o.get("xyz", "0 0 0")), rpy=_floats(o.get("rpy", "0 0 0")))


class Pose(_NS):
    pass


class Link(_NS):
    pass


class Joint(_NS):
    pass


class URDF:
    @classmethod
    def from_xml_file(cls, fn):
        root = ET.parse(fn).getroot()
        self = cls()
        self.name = root.get("name")
        self.links, self.joints = [], []
        for el in root:
            if el.tag == "link":
                vis = el.find("visual")
                visual = None
                if vis is not None:
                    mesh = vis.find("geometry/mesh")
                    geom = _NS(filename=mesh.get("filename") if mesh is not None else None)
                    visual = _NS(origin=_origin(vis), geometry=geom)
                self.links.append(Link(name=el.get("name"), visual=visual))
            elif el.tag == "joint":
                ax = el.find("axis")
                lim = el.find("limit")
                limit = None
                if lim is not None:
                    limit = _NS(
                        lower=float(lim.get("lower", 0.0)),
                        upper=float(lim.get("upper", 0.0)),
                        velocity=float(lim.get("velocity", 0.0)),
                    )
                self.joints.append(
                    Joint(
                        name=el.get("name"),
                        type=el.get("type"),
                        parent=el.find("parent").get("link"),
                        child=el.find("child").get("link"),
                        origin=_origin(el),
                        axis=_floats(ax.get("xyz")) if ax is not None else None,
                        limit=limit,
                    )
                )
        self.joint_map = {j.name: j for j in self.joints}
        self.link_map = {l.name: l for l in self.links}
        self._parent = {j.child: (j.name, j.parent) for j in self.joints}
        return self

    def get_root(self):
        children = set(self._parent)
        roots = [l.name for l in self.links if l.name not in children]
        assert len(roots) == 1
        return roots[0]

    def get_chain(self, root, tip, joints=True, links=True, fixed=True):
        chain = []
        link = tip
        while link != root:
            jn, parent = self._parent[link]
            if links:
                chain.append(link)
            if joints:
                chain.append(jn)
            link = parent
        if links:
            chain.append(root)
        chain.reverse()
        return chain


def _install_urdf():
    pkg = types.ModuleType("urdf_parser_py")
    mod = types.ModuleType("urdf_parser_py.urdf")
    mod.URDF, mod.Joint, mod.Link, mod.Pose = URDF, Joint, Link, Pose
    pkg.urdf = mod
    sys.modules["urdf_parser_py"] = pkg
    sys.modules["urdf_parser_py.urdf"] = mod


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install():
    """Install the stand-ins and load the reference modules by path. Returns a namespace."""
    _install_casadi()
    _install_urdf()
    for empty in ("xacro", "pyrender", "_init_paths"):
        sys.modules[empty] = types.ModuleType(empty)
    t3d = types.ModuleType("transforms3d")
    t3dq = types.ModuleType("transforms3d.quaternions")
    t3dq.quat2mat = t3dq.mat2quat = None
    t3d.quaternions = t3dq
    sys.modules["transforms3d"] = t3d
    sys.modules["transforms3d.quaternions"] = t3dq

    optas = types.ModuleType("optas")
    optas.__path__ = [REF + "/optas"]
    sys.modules["optas"] = optas
    vis = types.ModuleType("optas.visualize")
    vis.Visualizer = object
    sys.modules["optas.visualize"] = vis
    optas.visualize = vis

    out = types.SimpleNamespace()
    out.spatialmath = _load("optas.spatialmath", REF + "/optas/spatialmath.py")
    out.models = _load("optas.models", REF + "/optas/models.py")
    out.sdf_callback = _load("ref_sdf_callback", REF + "/gto/sdf_callback.py")
    out.utils = _load("ref_gto_utils", REF + "/gto/utils.py")
    out.dpc = _load("ref_depth_point_cloud", REF + "/mesh_to_sdf/depth_point_cloud.py")
    return out
