"""CPU: the N>1 sharding path with world_size 2 on gloo (the solver callable is the CPU oracle here;
on GPU ranks it is SolverHandle.solve_batch)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from grasptrajopt_amd.parallel import BatchPipeline, merge_batches, shard_by_scene, shard_range, split_results


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 64, 65, 16384):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_shard_by_scene_keeps_scenes_together():
    rng = np.random.default_rng(0)
    sid = rng.integers(0, 37, size=500)
    a = shard_by_scene(sid, 8)
    for s in np.unique(sid):
        assert len(set(a[sid == s])) == 1
    load = np.bincount(a, minlength=8)
    assert load.max() - load.min() <= np.bincount(sid).max()


def test_batch_pipeline_orders_results_and_reuses_handles():
    import threading
    import time

    class Fake:
        def __init__(self, tag):
            self.tag, self.busy, self.calls = tag, threading.Lock(), 0

        def solve_batch(self, x, delay):
            assert self.busy.acquire(blocking=False), "a handle was entered by two threads at once"
            try:
                time.sleep(delay)
                self.calls += 1
                return x * 2
            finally:
                self.busy.release()

    fakes = [Fake(i) for i in range(3)]
    with BatchPipeline(fakes) as pipe:
        t0 = time.perf_counter()
        out = pipe.solve_batches([(i, 0.05 if i % 2 else 0.01) for i in range(12)])
        el = time.perf_counter() - t0
    assert out == [2 * i for i in range(12)]
    assert sum(f.calls for f in fakes) == 12 and all(f.calls > 0 for f in fakes)
    assert el < 0.9 * (6 * 0.05 + 6 * 0.01)  # batches overlapped
    with pytest.raises(ValueError):
        BatchPipeline([])


def test_merged_batches_split_back_exactly(oracle_mod):
    """Folding several batches into one solve_batch call (ragged goal sets, scalar and per-instance
    arguments, different batch sizes) gives every batch exactly the result it gets alone: run on the oracle,
    whose solve_batch has the SolverHandle signature."""
    from helpers import Problem
    probs = [Problem("panda", B=3, scene_seed=1, n=32, res=0.07, n_goals=2), Problem("panda", B=2, scene_seed=1, n=32, res=0.07),
             Problem("panda", B=4, scene_seed=1, n=32, res=0.07, n_goals=3)]
    o = oracle_mod.Oracle(probs[0].desc, probs[0].cfg["link_ee"], probs[0].cfg["link_gripper"], oracle_mod.reference_opts(max_iter=3))
    for p in probs:
        p.finish(o.eval_fk)
    o.set_scene(*probs[0].scene_args())
    batches = [(0, p.qc, p.goals, p.n_goals, p.S, p.base, p.Q0) for p in probs]
    batches[1] = (np.zeros(2, np.int32), probs[1].qc, probs[1].goals, 1, np.tile(probs[1].S.reshape(1, 16), (2, 1)), probs[1].base[0], probs[1].Q0)
    merged, sizes = merge_batches(batches)
    assert sizes == [3, 2, 4] and merged[2].shape == (9, 3, 16) and merged[3].tolist() == [2] * 3 + [1] * 2 + [3] * 4
    alone = [o.solve_batch(*b, n_threads=1) for b in batches]

    class Lane:
        def solve_batch(self, *a):
            return o.solve_batch(*a, n_threads=1)

    with BatchPipeline([Lane(), Lane()]) as pipe:
        for merge in (1, 2, 3):
            got = pipe.solve_batches(batches, merge=merge)
            assert len(got) == 3
            for a, b in zip(alone, got):
                for x, y in zip(a, b):
                    np.testing.assert_array_equal(x, y)
    parts = split_results(o.solve_batch(*merged, n_threads=1), sizes)
    np.testing.assert_array_equal(parts[2][0], alone[2][0])
    with pytest.raises(ValueError):
        merge_batches([batches[0], batches[1][:4] + (None,) + batches[1][5:]])


def test_batch_pipeline_propagates_errors():
    class Bad:
        def solve_batch(self):
            raise RuntimeError("boom")

    with BatchPipeline([Bad()]) as pipe:
        with pytest.raises(RuntimeError):
            pipe.submit("solve_batch").result()
        # the handle went back to the idle queue
        with pytest.raises(RuntimeError):
            pipe.submit("solve_batch").result()


def _worker(rank, world, port, tmp):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from helpers import Problem
    from grasptrajopt_amd.parallel import solve_sharded
    from oracle import oracle
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    prob = Problem("panda", B=5, scene_seed=1, n=32, res=0.07)
    o = oracle.Oracle(prob.desc, prob.cfg["link_ee"], prob.cfg["link_gripper"], oracle.reference_opts(max_iter=4))
    prob.finish(o.eval_fk)
    o.set_scene(*prob.scene_args())
    solve = lambda *a: o.solve_batch(*a, n_threads=1)
    idx, Q, dQ, cost, iters, status = solve_sharded(solve, np.zeros(5, np.int32), prob.qc, prob.goals, 1, prob.S,
                                                    prob.base, prob.Q0, rank, world)
    np.savez(os.path.join(tmp, f"r{rank}.npz"), idx=idx, Q=Q, cost=cost, iters=iters)
    if rank == 0:
        Qs, _, cs, its, _ = o.solve_batch(0, prob.qc, prob.goals, 1, prob.S, prob.base, prob.Q0, n_threads=1)
        np.savez(os.path.join(tmp, "single.npz"), Q=Qs, cost=cs, iters=its)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_solve_equals_single_process(tmp_path, oracle_mod):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = np.load(tmp_path / "single.npz")
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        assert z["idx"].tolist() == list(range(5))
        np.testing.assert_array_equal(z["Q"], single["Q"])      # same code on every rank: bit-identical
        np.testing.assert_array_equal(z["cost"], single["cost"])
        np.testing.assert_array_equal(z["iters"], single["iters"])


def test_bench_call_plan_times_exactly_k_steps():
    """bench.py splits its K timed steps into solver calls: never more than `merge` batches per call, exactly K in
    total, sizes within one of each other, a multiple of the lane count when K allows."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("bench", pathlib.Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for lanes in (1, 2, 4):
        for merge in (1, 4, 32):
            for n in list(range(0, 70)) + [96, 127, 128, 129, 512, 1000]:
                plan = bench.plan_calls(n, lanes, merge)
                assert sum(plan) == n and all(1 <= m <= merge for m in plan)
                if plan:
                    assert max(plan) - min(plan) <= 1
                    if n >= lanes * merge:
                        assert len(plan) % lanes == 0 or len(plan) == -(-n // merge)
    assert bench.plan_calls(512, 4, 32) == [32] * 16
    assert bench.plan_calls(5, 4, 32) == [2, 1, 1, 1]


def test_bench_final_line_stays_under_six_kilobytes():
    """The driver parses the LAST stdout line of bench.py and its record keeps an 8 KB tail: round 5's 21.6 KB line came back
    `parsed: null`.  bench.compact_line() of the largest lines on record (profiles/r05_bench*.json, 21 KB each) has to
    stay under 6000 bytes and carry every key of the contract, with the roofline recomputable from it."""
    import importlib.util
    import json
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("bench", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name in ("r05_bench_steps20.json", "r05_bench.json"):
        full = json.loads((root / "profiles" / name).read_text().strip().splitlines()[-1])
        assert len(json.dumps(full)) > 20000
        line = bench.compact_line(full)
        text = json.dumps(line)
        assert len(text) < bench.LINE_LIMIT <= 6000, len(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "timed_regions", "host_api", "quality", "other_configs", "lib_sha16"):
            assert k in line, k
        assert line["value"] == full["value"] and line["config"]["workload"].startswith("BASELINE configs[1]")
        rf = line["roofline"]
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "avg_launch_us",
                  "alg_bytes_skipped_frac", "traffic_over_alg_bytes", "dominant_by_time"):
            assert k in rf, k
        # recomputable: algorithmic bytes per launch over the launch's duration, over the peak
        assert abs(rf["alg_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9 / rf["peak"] - rf["frac"]) < 2e-3 * max(rf["frac"], 1e-3) + 1e-4
        cb = line["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
        assert line["quality"]["gate"] == "pass" and 0 < line["quality"]["goal_ok_frac"] <= 1
        for v in line["other_configs"].values():
            assert v["oracle_check_ok"] and v["trajectories_per_s"] > 0 and v["frac"] > 0
    # a line that would still be too long drops its optional objects instead of growing
    fat = dict(full, other_configs={f"configs[{i}]": v for i in range(40) for v in full["other_configs"].values()})
    fat["config"] = dict(fat["config"], workload=fat["config"]["workload"] + "x" * 5000)
    slim = bench.compact_line(fat)
    assert len(json.dumps(slim)) < 6000 and slim["other_configs"] is None and slim["roofline"]["frac"] == full["roofline"]["frac"]


def _wq_worker(rank, world, port, tmp):
    """Three ranks over gloo: (1) a world=1 solve inside the 3-rank default group keeps its local result (ADVICE round 5:
    it used to enter a 3-rank all_gather with a one-rank receive list); (2) the work queue against the static partition."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time
    import torch.distributed as dist
    from grasptrajopt_amd.parallel import scene_chunks, shard_by_scene, solve_local_shard, solve_sharded, solve_work_queue
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    B, ndof, T = 41, 3, 6
    rng = np.random.default_rng(5)
    sid = np.sort(rng.integers(0, 9, size=B)).astype(np.int32)
    qc, goals = rng.normal(size=(B, ndof)), rng.normal(size=(B, 1, 16))
    Q0 = rng.normal(size=(B, ndof, T))
    so, base = rng.normal(size=(B, 16)), rng.normal(size=(B, 3))
    calls = []

    def fake(scene_id, qc_, goals_, n_goals, standoff, base_pos, Q0_):
        # a deterministic function of an instance's own inputs; "difficulty" = time, uneven across scenes
        n = len(qc_)
        calls.append(n)
        time.sleep(0.002 * float(np.sum(np.asarray(scene_id) % 3 == 0)))
        Q = Q0_ * 2 + qc_[:, :, None] + np.asarray(goals_).reshape(n, -1)[:, :1, None] + np.asarray(standoff)[:, :1, None] + np.asarray(base_pos)[:, :1, None]
        return Q, Q[:, :, 1:] - Q[:, :, :-1], Q.sum(axis=(1, 2)), (np.asarray(scene_id) + 3).astype(np.int32), (np.asarray(scene_id) % 2).astype(np.int32)

    # (1) world = 1 with a 3-rank default group and no group passed: local result, no collective
    mine = np.arange(4)
    out = solve_local_shard(fake, mine, sid[mine], qc[mine], goals[mine], 1, so[mine], base[mine], Q0[mine], B=4, rank=0, world=1)
    assert np.array_equal(out[0], mine) and out[1].shape == (4, ndof, T)
    dist.barrier()
    # (2) static partition by scene, then the queue: same instances, any order of claiming, equal results
    owner = shard_by_scene(sid, world)
    ref = solve_sharded(fake, sid, qc, goals, 1, so, base, Q0, rank, world, assignment=owner)
    chunks = scene_chunks(sid, max_instances=4)
    assert sorted(np.concatenate(chunks).tolist()) == list(range(B)) and all(len(set(sid[c].tolist())) == 1 and len(c) <= 4 for c in chunks)
    make = lambda ix: (sid[ix], qc[ix], goals[ix], 1, so[ix], base[ix], Q0[ix])
    for fns in (fake, [fake, fake]):  # one claiming thread, two
        st = {}
        got = solve_work_queue(fns, chunks, make, B, rank, world, stats=st)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        assert sum(st["instances_by_rank"]) == B and st["chunks_total"] == len(chunks)
    # a rank that comes late finds the queue empty: it claims nothing and still takes part in the gather
    if rank == 2:
        time.sleep(1.5)
    st = {}
    got = solve_work_queue(fake, chunks, make, B, rank, world, stats=st)
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)
    assert st["instances_by_rank"][2] == 0 and sum(st["instances_by_rank"]) == B
    own = solve_work_queue(fake, chunks, make, B, rank, world, gather=False)
    assert np.all(np.diff(own[0]) > 0)
    np.testing.assert_array_equal(own[1], ref[1][own[0]])
    np.savez(os.path.join(tmp, f"wq{rank}.npz"), idx=own[0])
    # a rank whose solver fails makes every rank raise instead of leaving the others in the gather
    def bad(*a):
        if rank == 1:
            raise ValueError("boom")
        return fake(*a)
    try:
        solve_work_queue(bad, chunks, make, B, rank, world)
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    dist.barrier()
    dist.destroy_process_group()


def test_work_queue_three_ranks_equals_static_partition(tmp_path):
    """Balance across ranks: scene-grouped chunks claimed through the process
    group's store counter; three ranks over gloo, every instance solved exactly once, results equal to the static partition's."""
    port = 31500 + os.getpid() % 2000
    mp.spawn(_wq_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    idx = [np.load(tmp_path / f"wq{r}.npz")["idx"] for r in range(3)]
    allidx = np.concatenate(idx)
    assert sorted(allidx.tolist()) == list(range(41))  # a partition: nothing twice, nothing missing


def test_work_queue_single_process_without_a_group():
    from grasptrajopt_amd.parallel import scene_chunks, solve_work_queue
    sid = np.array([2, 2, 0, 0, 0, 7], dtype=np.int32)
    chunks = scene_chunks(sid)
    assert [c.tolist() for c in chunks] == [[0, 1], [2, 3, 4], [5]]
    Q0 = np.arange(6 * 2 * 3, dtype=np.float64).reshape(6, 2, 3)
    f = lambda s, q, g, n, so, b, Q: (Q + 1, Q[:, :, 1:], Q.sum(axis=(1, 2)), np.asarray(s, np.int32), np.zeros(len(Q), np.int32))
    make = lambda ix: (sid[ix], None, None, 1, None, None, Q0[ix])
    idx, Q, dQ, c, it, st = solve_work_queue([f, f, f], chunks, make, 6, 0, 1)
    assert idx.tolist() == list(range(6))
    np.testing.assert_array_equal(Q, Q0 + 1)
    np.testing.assert_array_equal(it, sid)


def test_bench_lanes_per_rank_fit_the_cgroup_quota():
    """bench.py, N > 1: a lane is a host thread of 0.27 cores (measured), a rank's main thread 0.5: four lanes at the 16 CPUs
    the driver box's container showed for eight ranks (round 5 cut to quota // ranks = 2), fewer when the quota is smaller,
    never none, and whatever was asked for without a quota or with one rank."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("bench", pathlib.Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.lanes_that_fit(4, 8, 16.0)[0] == 4
    assert bench.lanes_that_fit(4, 8, 12.0)[0] == 3
    assert bench.lanes_that_fit(4, 8, 8.0)[0] == 1
    assert bench.lanes_that_fit(4, 8, 2.0)[0] == 1
    assert bench.lanes_that_fit(4, 8, None)[0] == 4 and bench.lanes_that_fit(4, 1, 1.0)[0] == 4
    assert bench.lanes_that_fit(4, 2, 16.0)[0] == 4 and bench.lanes_that_fit(6, 4, 8.0)[0] == 5
    for asked, world, quota in ((4, 8, 16.0), (4, 8, 9.0), (2, 4, 3.0)):
        lanes, why = bench.lanes_that_fit(asked, world, quota)
        assert 1 <= lanes <= asked and str(lanes) in why
        assert lanes == 1 or world * (bench.RANK_CORES + lanes * bench.LANE_CORES) <= quota + 1e-9
