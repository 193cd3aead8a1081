"""Triangle-mesh reading (OBJ / STL) and seeded, area-weighted surface sampling.

Replaces the reference's ``trimesh.load`` + ``mesh_to_sdf.get_surface_point_cloud(mesh, 'sample',
sample_point_count=100)`` (gto/gto_models.py:75-77, mesh_to_sdf/surface_point_cloud.py:177-188).
The reference draws these points from an unseeded RNG, so they are an *input* of the path and not
reproducible; here the draw is seeded so fixtures are deterministic.
"""
from __future__ import annotations

import os
import struct
from typing import Tuple

import numpy as np


def _load_obj(path: str) -> Tuple[np.ndarray, np.ndarray]:
    verts, faces = [], []
    with open(path, "r", errors="ignore") as fh:
        for line in fh:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):  # fan triangulation
                    faces.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


def _load_stl(path: str) -> Tuple[np.ndarray, np.ndarray]:
    with open(path, "rb") as fh:
        data = fh.read()
    ntri = struct.unpack_from("<I", data, 80)[0] if len(data) >= 84 else 0
    if len(data) == 84 + 50 * ntri:  # binary
        rec = np.frombuffer(data, dtype=np.uint8, offset=84).reshape(ntri, 50)
        tri = rec[:, 12:48].copy().view("<f4").reshape(ntri, 3, 3).astype(np.float64)
    else:  # ascii
        pts = []
        for line in data.decode("ascii", errors="ignore").splitlines():
            s = line.strip()
            if s.startswith("vertex"):
                pts.append([float(x) for x in s.split()[1:4]])
        tri = np.asarray(pts, dtype=np.float64).reshape(-1, 3, 3)
    verts = tri.reshape(-1, 3)
    faces = np.arange(verts.shape[0], dtype=np.int64).reshape(-1, 3)
    return verts, faces


def load_mesh(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Return (vertices (V,3) f64, triangles (F,3) i64)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return _load_obj(path)
    if ext == ".stl":
        return _load_stl(path)
    raise ValueError(f"unsupported mesh format '{ext}' ({path}); convert to OBJ or STL")


def sample_surface(verts: np.ndarray, faces: np.ndarray, count: int, seed: int = 0):
    """Area-weighted uniform surface samples. Returns (points (count,3), normals (count,3))."""
    a = verts[faces[:, 0]]
    e1 = verts[faces[:, 1]] - a
    e2 = verts[faces[:, 2]] - a
    cross = np.cross(e1, e2)
    area2 = np.linalg.norm(cross, axis=1)
    cdf = np.cumsum(area2)
    if cdf[-1] <= 0:
        raise ValueError("mesh has zero surface area")
    rng = np.random.default_rng(seed)
    fi = np.searchsorted(cdf, rng.random(count) * cdf[-1], side="right")
    fi = np.minimum(fi, len(faces) - 1)
    u = rng.random(count)
    v = rng.random(count)
    flip = u + v > 1.0
    u = np.where(flip, 1.0 - u, u)
    v = np.where(flip, 1.0 - v, v)
    pts = a[fi] + u[:, None] * e1[fi] + v[:, None] * e2[fi]
    nrm = cross[fi] / np.maximum(area2[fi], 1e-300)[:, None]
    return pts, nrm
