"""GPU: the multi-rank path.  One GPU box has one device, so the ranks of these tests share it: with gloo as the backend
of the (single, out-of-the-solve) collective the whole N > 1 path runs with the HIP solver on every rank -- scene-grouped
shards, one SolverHandle per rank, all_gather of the results -- and must reproduce a single-process solve bit for bit.
The RCCL smoke test needs one device per rank: RCCL refuses two ranks on one device, in which case that test is skipped
with RCCL's message (the 8-GPU run is the driver's)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    sys.path.insert(0, os.environ["GTO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GTO_ROOT"], "tests"))
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from grasptrajopt_amd import _capi, synthetic as syn
    from grasptrajopt_amd.parallel import shard_by_scene, solve_sharded
    from helpers import Problem
    from oracle import oracle
    prob = Problem("panda", B=12, scene_seed=2)
    opts = oracle.reference_opts(max_iter=10)
    h = _capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    prob.finish(h.eval_fk)
    sid = np.array([0, 1] * 6, dtype=np.int32)
    owner = shard_by_scene(sid, world)
    for s in np.unique(sid[owner == rank]):
        h.set_scene(int(s), *prob.scene_args()[1:])
    idx, Q, dQ, f, it, st = solve_sharded(h.solve_batch, sid, prob.qc, prob.goals, 1, prob.S, prob.base, prob.Q0, rank=rank,
                                          world=world, assignment=owner)
    if rank == 0:
        h.set_scene(0, *prob.scene_args()[1:]); h.set_scene(1, *prob.scene_args()[1:])
        ref = h.solve_batch(sid, prob.qc, prob.goals, 1, prob.S, prob.base, prob.Q0)
        print("RESULT", json.dumps({"n": int(len(idx)), "equal": bool(np.array_equal(Q, ref[0]) and np.array_equal(it, ref[3]))}))
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_ranks_all_gather_over_rccl(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, GTO_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", str(script)]
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.skip("two RCCL ranks on one device did not finish within 240 s")
    out = res.stdout + res.stderr
    if res.returncode != 0:
        if any(k in out for k in ("Duplicate GPU", "duplicate GPU", "invalid usage", "ncclInvalidUsage", "NCCL error", "ncclUnhandledCudaError")):
            pytest.skip("RCCL refuses two ranks on one device: " + out.strip().splitlines()[-1][:200])
        raise AssertionError(out[-3000:])
    line = [l for l in out.splitlines() if l.startswith("RESULT")][0]
    import json
    r = json.loads(line[7:])
    assert r["n"] == 12 and r["equal"]


def test_one_rank_over_rccl(tmp_path):
    """The RCCL branch executed on the hardware that is there: the same worker as the two-rank test with ONE rank -- process
    group over `nccl` bound to the device, the error-flag all_reduce, the two all_gathers of the result payloads on device
    tensors, the barrier.  (Two ranks need two devices; this is as far as one device goes.)"""
    script = tmp_path / "worker1.py"
    script.write_text(WORKER)
    env = dict(os.environ, GTO_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29733", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-3000:]
    import json
    r = json.loads([l for l in out.splitlines() if l.startswith("RESULT")][0][7:])
    assert r["n"] == 12 and r["equal"]


WORKER_GLOO = textwrap.dedent('''
    import os, sys, json
    import numpy as np
    sys.path.insert(0, os.environ["GTO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GTO_ROOT"], "tests"))
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grasptrajopt_amd import _capi, synthetic as syn
    from grasptrajopt_amd.parallel import shard_by_scene, solve_sharded
    from helpers import Problem
    from oracle import oracle
    n_sc, SG = 16, 8
    B = n_sc * SG
    prob = Problem("panda", B=B, scene_seed=2, n=40, res=0.056)
    opts = oracle.reference_opts(max_iter=14)
    h = _capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)   # this rank's own handle
    prob.finish(h.eval_fk)
    scenes = [syn.make_scene(50 + s, n=40, res=0.056) for s in range(n_sc)]
    sid = np.repeat(np.arange(n_sc, dtype=np.int32), SG)
    owner = shard_by_scene(sid, world)
    for s in np.unique(sid[owner == rank]):          # a scene's field lives on exactly one rank
        sc = scenes[int(s)]
        h.set_scene(int(s), sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
    idx, Q, dQ, f, it, st = solve_sharded(h.solve_batch, sid, prob.qc, prob.goals, 1, prob.S, prob.base, prob.Q0, rank=rank,
                                          world=world, assignment=owner)
    scenes_here = int(len(np.unique(sid[owner == rank])))
    # the same batch through the work queue: chunks = scenes, claimed through the group's store counter; the rank that
    # draws a chunk uploads its scene (two claiming handles on this rank: two chunks in flight)
    from grasptrajopt_amd.parallel import scene_chunks, solve_work_queue
    h2 = _capi.SolverHandle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], opts, device=0)
    S16 = np.broadcast_to(np.asarray(prob.S, dtype=np.float64).reshape(-1, 16), (B, 16))
    def claimer(hh):
        def solve(ix_sid, *a):
            sc = scenes[int(ix_sid[0])]
            hh.set_scene(int(ix_sid[0]), sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
            return hh.solve_batch(ix_sid, *a)
        return solve
    make = lambda ix: (sid[ix], prob.qc[ix], prob.goals[ix], 1, S16[ix], prob.base[ix], prob.Q0[ix])
    wst = {}
    wq = solve_work_queue([claimer(h), claimer(h2)], scene_chunks(sid), make, B, rank, world, stats=wst)
    wq_eq = all(bool(np.array_equal(a, b)) for a, b in zip((idx, Q, dQ, f, it, st), wq))
    if rank == 0:
        for s in range(n_sc):
            sc = scenes[s]
            h.set_scene(s, sc.c_all, sc.c_obs, sc.shape, sc.origin, sc.res)
        ref = h.solve_batch(sid, prob.qc, prob.goals, 1, prob.S, prob.base, prob.Q0)
        eq = all(bool(np.array_equal(a, b)) for a, b in zip((Q, dQ, f, it, st), ref))
        print("RESULT", json.dumps({"n": int(len(idx)), "equal": eq, "scenes_on_rank0": scenes_here, "iters_distinct": int(len(set(it.tolist()))),
                                    "world": world, "work_queue_equal": wq_eq, "work_queue_by_rank": wst["instances_by_rank"]}))
    h2.close()
    dist.barrier()
    h.close()
    dist.destroy_process_group()
''')


def _run_ranks(script, nproc, port, timeout=600, extra_env=None):
    env = dict(os.environ, GTO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_one_device_hip_solver_equals_single_process(tmp_path, world):
    """The N > 1 path with the HIP solver on every rank: `world` processes share the one device, gloo carries the single
    all_gather, 16 scenes x 8 grasps are grouped by scene onto the ranks (every field on exactly one rank, its own handle);
    every rank's gathered result equals rank 0's single-process solve of all 128 instances bit for bit."""
    script = tmp_path / "worker_gloo.py"
    script.write_text(WORKER_GLOO)
    res = _run_ranks(script, world, 29741 + world)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-3000:]
    import json
    r = json.loads([l for l in out.splitlines() if l.startswith("RESULT")][0][7:])
    assert r["world"] == world and r["n"] == 128 and r["equal"]
    assert 0 < r["scenes_on_rank0"] < 16 and r["iters_distinct"] > 1
    # the work queue (scene chunks claimed through the store counter, two claiming handles per rank): same bits, every instance once
    assert r["work_queue_equal"] and sum(r["work_queue_by_rank"]) == 128 and len(r["work_queue_by_rank"]) == world


def _bench_lines(stdout):
    """bench.py prints the full record on a `BENCH_DETAIL` line and the compact line the driver parses LAST (under 6 KB,
    with the contract's keys); the tests read the detail and require the compact line to agree on the headline."""
    import json
    lines = stdout.splitlines()
    compact = json.loads([l for l in lines if l.startswith("{")][-1])
    assert lines[-1].startswith("{") and len(lines[-1]) < 6000, len(lines[-1])
    detail = json.loads([l for l in lines if l.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "lib_sha16"):
        assert compact[k] == detail[k], k
    assert compact["roofline"]["frac"] == detail["roofline"]["frac"] and compact["quality"]["gate"] == detail["quality"]["gate"]
    return detail


def test_bench_two_ranks_dry_run_on_one_device():
    """bench.py's N > 1 code path (rank environment, barrier + max-over-ranks timing, scene-sharded leg with its gather),
    run as the driver would launch it but with gloo and both ranks on the one device: the JSON line says n_gpus 2, the
    scene-sharded leg returned every instance in order, the lanes' results are reproducible."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "8", "--warmup", "1",
           "--merge", "2", "--repeats", "2", "--scenes-per-gpu", "6", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    d = _bench_lines(res.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "weak"
    ss = d["scene_sharded"]
    assert ss["instances"] == 2 * 6 * 8 and ss["all_instances_returned"] and ss["own_shard_round_trip_exact"]
    assert ss["max_joint_limit_violation"] <= 1e-8
    assert d["pipeline"]["lane_results_reproducible_alone"] and d["quality"]["gate"] == "pass"


def test_bench_eight_ranks_dry_run_on_one_device():
    """The 8-rank code path without an 8-GPU box: bench.py as the driver launches it for N = 8, with
    gloo and all eight ranks on the one device: spawn path, shard_by_scene over eight ranks (4 scenes each), the gather of
    eight shards with every instance back in order and the rank's own shard bit-exact through the gather, the `collective`
    object with eight rows and the lane decision (lanes per rank = what fits the cgroup quota at the measured cores per lane)."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--same-device", "--steps", "4", "--warmup", "1",
           "--merge", "1", "--repeats", "1", "--scenes-per-gpu", "4", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    d = _bench_lines(res.stdout)
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["scaling"] == "weak"
    ss = d["scene_sharded"]
    assert ss["instances"] == 8 * 4 * 8 and ss["all_instances_returned"] and ss["own_shard_round_trip_exact"]
    assert ss["max_joint_limit_violation"] <= 1e-8
    col = d["collective"]
    assert col["world_size"] == 8 and len(col["ranks"]) == 8 and sorted(r["rank"] for r in col["ranks"]) == list(range(8))
    assert 1 <= col["lanes_per_rank"] <= 4 and col["lane_decision"]
    q = col["cgroup_cpu_quota_cores"]
    if q is not None:  # lanes that fit the quota at the measured 0.27 cores per lane + 0.5 per rank: four at quota 16 / 8 ranks
        assert col["lanes_per_rank"] == max(1, min(4, int((q / 8 - 0.5) / 0.27 + 1e-9)))
    assert col["host_cpu_cores_busy_all_ranks"] > 0
    assert d["quality"]["gate"] == "pass"


def test_bench_rccl_branch_with_one_rank():
    """bench.py's `nccl` branch on the one device there is: launched as the driver launches the N > 1 runs (torch.distributed.run),
    with one rank and --force-dist: process group over RCCL bound to the device, barriers, the all_reduce of the timings on
    device tensors, the scene-sharded leg's all_gather pair on the device, the `collective` object with RCCL's version."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29747",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl", "--steps", "4", "--warmup", "1", "--merge", "1",
           "--repeats", "1", "--scenes-per-gpu", "4", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    d = _bench_lines(res.stdout)
    col = d["collective"]
    assert col["backend"] == "nccl" and col["is_rccl"] and col["world_size"] == 1 and col["nccl_version"]
    assert col["ranks"][0]["all_gather"]["backend"] == "nccl" and col["ranks"][0]["all_gather"]["collectives"] == 2
    ss = d["scene_sharded"]
    assert ss["instances"] == 4 * 8 and ss["all_instances_returned"] and ss["own_shard_round_trip_exact"]
    assert d["quality"]["gate"] == "pass"
