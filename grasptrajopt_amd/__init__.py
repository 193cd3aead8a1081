"""grasptrajopt_amd — MI355X-native inner solver for Grasping Trajectory Optimization (GTO).

Only the hot path of IRVLUTD/GraspTrajOpt lives here: the trajectory solve behind
GTOPlanner.plan()/plan_goalset().  Public surface:
    GTORobotModel, GTOPlanner            drop-in for the reference's gto.gto_models / gto.gto_planner
    IKSolver                             drop-in for gto.ik_solver (batched on the GPU)
    DepthPointCloud                      drop-in for mesh_to_sdf.depth_point_cloud (cost fields on the GPU)
    BasePlanner                          drop-in for gto.base_planner (mobile base placement on the GPU)
    optas_facade                         OptimizationBuilder / CasADiSolver-shaped recorder + solver
    _capi.SolverHandle                   thin ctypes binding of the C ABI (include/gto_solver.h)
"""
from .gto_models import GTORobotModel  # noqa: F401
from .gto_planner import GTOPlanner  # noqa: F401
from .ik_solver import IKSolver  # noqa: F401
from .depth_point_cloud import DepthPointCloud  # noqa: F401
from .base_planner import BasePlanner  # noqa: F401
from .robot_desc import RobotDesc, load_builtin  # noqa: F401

__all__ = ["GTORobotModel", "GTOPlanner", "IKSolver", "DepthPointCloud", "BasePlanner", "RobotDesc", "load_builtin"]
