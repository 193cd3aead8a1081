#!/usr/bin/env python3
"""tests/golden/plan_statistics.npz: per-plan shape statistics of the reference's stored GTO plans
(/root/reference/examples/results_iros2024/GTO_scenereplica_{panda,fetch}_tabletop_*.json: outputs of the reference's own
CasADi / IPOPT path on SceneReplica inputs that are not in the repository).  Data only: eight numbers per plan
(grasptrajopt_amd.results.plan_shape_statistics), no plan and no reference text is stored.
Run in the container that has /root/reference:  python tests/golden/make_plan_statistics.py"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from grasptrajopt_amd.results import plan_shape_statistics  # noqa: E402
from grasptrajopt_amd.robot_desc import load_builtin  # noqa: E402

REF = "/root/reference"
out = {}
for tag, robot in (("panda_tabletop", "panda"), ("fetch_tabletop", "fetch")):
    fn = glob.glob(f"{REF}/examples/results_iros2024/GTO_scenereplica_{tag}_24-*.json")[0]
    plans = []
    for scene in json.load(open(fn)).values():
        for order in scene.values():
            for obj in order.values():
                if isinstance(obj, dict) and obj.get("plan") is not None:
                    p = np.array(obj["plan"], dtype=np.float64)
                    if p.ndim == 2 and p.shape[1] == 50:
                        plans.append(p)
    d = load_builtin(robot)
    st = plan_shape_statistics(np.stack(plans), d.opt_index, d.lower[d.opt_index], d.upper[d.opt_index])
    for k, v in st.items():
        out[f"{tag}/{k}"] = v
    print(tag, len(plans), {k: np.round(np.percentile(v, [5, 50, 95]), 4).tolist() for k, v in st.items()})
np.savez_compressed(os.path.join(HERE, "plan_statistics.npz"), **out)
