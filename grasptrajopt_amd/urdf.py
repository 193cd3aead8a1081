"""Minimal URDF reader (ElementTree) — the kinematic facts the GTO path needs.

The reference gets these through ``urdf_parser_py`` (optas/models.py:276-293); that package is not
a dependency here.  Semantics kept identical to what optas reads from it:
  * joints/links keep document order (actuated-joint order = order of non-fixed joints,
    optas/models.py:349-354),
  * a joint without <origin> has zero xyz/rpy (optas/models.py:642-651), without <axis> the axis is
    (1,0,0) (optas/models.py:653-659), without <limit> the limits are -1e9/+1e9
    (optas/models.py:438-456); a <limit> lacking lower/upper yields 0 (README.md:120-121),
  * a link's visual is its FIRST <visual>; visual origin defaults to zero (optas/models.py:629-640).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional


def _floats(s: str) -> List[float]:
    return [float(x) for x in s.split()]


@dataclass
class UrdfLink:
    name: str
    has_visual: bool = False
    visual_xyz: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    visual_rpy: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    visual_mesh: Optional[str] = None
    visual_scale: List[float] = field(default_factory=lambda: [1.0, 1.0, 1.0])


@dataclass
class UrdfJoint:
    name: str
    type: str
    parent: str
    child: str
    xyz: List[float]
    rpy: List[float]
    axis: List[float]
    lower: float
    upper: float
    velocity: float


class Urdf:
    def __init__(self, root: ET.Element):
        self.name: str = root.get("name")
        self.links: List[UrdfLink] = []
        self.joints: List[UrdfJoint] = []
        for el in root:
            if el.tag == "link":
                self.links.append(self._parse_link(el))
            elif el.tag == "joint":
                self.joints.append(self._parse_joint(el))
        self.link_map: Dict[str, UrdfLink] = {l.name: l for l in self.links}
        self.joint_map: Dict[str, UrdfJoint] = {j.name: j for j in self.joints}
        self.parent_joint: Dict[str, UrdfJoint] = {j.child: j for j in self.joints}

    @classmethod
    def from_file(cls, filename: str) -> "Urdf":
        return cls(ET.parse(filename).getroot())

    @classmethod
    def from_string(cls, text: str) -> "Urdf":
        return cls(ET.fromstring(text))

    @staticmethod
    def _parse_link(el: ET.Element) -> UrdfLink:
        link = UrdfLink(name=el.get("name"))
        vis = el.find("visual")
        if vis is not None:
            link.has_visual = True
            o = vis.find("origin")
            if o is not None:
                link.visual_xyz = _floats(o.get("xyz", "0 0 0"))
                link.visual_rpy = _floats(o.get("rpy", "0 0 0"))
            mesh = vis.find("geometry/mesh")
            if mesh is not None:
                link.visual_mesh = mesh.get("filename")
                if mesh.get("scale"):
                    link.visual_scale = _floats(mesh.get("scale"))
        return link

    @staticmethod
    def _parse_joint(el: ET.Element) -> UrdfJoint:
        o = el.find("origin")
        xyz = _floats(o.get("xyz", "0 0 0")) if o is not None else [0.0, 0.0, 0.0]
        rpy = _floats(o.get("rpy", "0 0 0")) if o is not None else [0.0, 0.0, 0.0]
        ax = el.find("axis")
        axis = _floats(ax.get("xyz")) if ax is not None else [1.0, 0.0, 0.0]
        lim = el.find("limit")
        if lim is None:
            lower, upper, velocity = -1e9, 1e9, 1e9
        else:
            lower = float(lim.get("lower", 0.0))
            upper = float(lim.get("upper", 0.0))
            velocity = float(lim.get("velocity", 0.0))
        return UrdfJoint(
            name=el.get("name"),
            type=el.get("type"),
            parent=el.find("parent").get("link"),
            child=el.find("child").get("link"),
            xyz=xyz,
            rpy=rpy,
            axis=axis,
            lower=lower,
            upper=upper,
            velocity=velocity,
        )

    def get_root(self) -> str:
        roots = [l.name for l in self.links if l.name not in self.parent_joint]
        if len(roots) != 1:
            raise ValueError(f"URDF must have exactly one root link, found {roots}")
        return roots[0]

    def get_chain(self, root: str, tip: str) -> List[str]:
        """Joint names from ``root`` down to ``tip`` (urdf_parser_py get_chain(links=False))."""
        chain: List[str] = []
        link = tip
        while link != root:
            j = self.parent_joint[link]
            chain.append(j.name)
            link = j.parent
        chain.reverse()
        return chain
