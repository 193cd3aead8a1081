"""Multi-GPU layer: one process per GPU, instances sharded statically (shard_range / shard_by_scene) or drawn from a
shared work queue (solve_work_queue), no data-path collective.

Every (scene, goal-set) instance is an independent problem (gto/gto_planner.py:185-245 shares
nothing across calls), so a node's 8 GPUs each solve a contiguous block of the instance list,
grouped by scene so that a scene's cost field is uploaded to exactly one GPU.
The only communication is an optional all_gather of the solved trajectories (~3.6 KB per
instance) over RCCL/xGMI (`backend="nccl"` on ROCm) or gloo on CPU-only test boxes.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition [lo, hi) of n_items over `world` ranks (sizes differ by <= 1)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_scene(scene_id: Sequence[int], world: int) -> np.ndarray:
    """Assign instances to ranks so that all instances of one scene land on the same rank and the
    per-rank instance counts are balanced greedily (largest scene first). Returns rank per instance."""
    scene_id = np.asarray(scene_id)
    scenes, counts = np.unique(scene_id, return_counts=True)
    load = np.zeros(world, dtype=np.int64)
    owner: Dict[int, int] = {}
    for s, c in sorted(zip(scenes.tolist(), counts.tolist()), key=lambda sc: (-sc[1], sc[0])):
        r = int(np.argmin(load))
        owner[s] = r
        load[r] += c
    return np.array([owner[int(s)] for s in scene_id], dtype=np.int32)


def _shard_indices(B: int, world: int, assignment: Optional[np.ndarray]):
    """Global instance indices of every rank's shard (ascending inside a shard)."""
    if assignment is None:
        return [np.arange(*shard_range(B, r, world)) for r in range(world)]
    assignment = np.asarray(assignment)
    return [np.nonzero(assignment == r)[0] for r in range(world)]


def _group_exists() -> bool:
    """A torch.distributed process group has been initialised in this process (a one-rank group still runs the collective:
    the RCCL branch can be exercised on a one-GPU box)."""
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def _collective_applies(world: int, group=None) -> bool:
    """Whether a solve over `world` ranks ends in the collective.  world > 1: always (the caller's `group` has to have that
    many ranks).  world == 1: only when the group the collective would run on HAS exactly one rank -- the one-rank RCCL
    branch on a one-GPU box; a world=1 solve inside a larger job (default group of N > 1 ranks, no group passed) keeps its
    local result instead of entering an N-rank collective with a one-rank receive list."""
    if world > 1:
        return True
    if not _group_exists():
        return False
    import torch.distributed as dist
    try:
        return dist.get_world_size(group) == 1
    except Exception:
        return False


def gather_results(mine: np.ndarray, result: tuple, B: int, rank: int, world: int, group=None,
                   assignment: Optional[np.ndarray] = None, stats: Optional[dict] = None):
    """all_gather the solved shards (Q, dQ, cost, iters, status of the instances `mine`) so that every rank holds the
    full batch of B instances in the original order: the ONLY collective of the multi-GPU path, outside the solve.
    Two fixed-size payloads per rank, padded to the largest shard: float64 (Q, dQ, cost) and int32 (iterations, status:
    integers travel as integers); on the device for RCCL (`backend="nccl"`), on the host for gloo.  A rank whose `mine`
    does not match the assignment makes EVERY rank raise (the error flag is reduced first: nobody is left waiting in
    the collective).  `stats`, if given, receives the bytes this rank sent and received and the seconds the collectives took."""
    import time
    import torch
    import torch.distributed as dist
    Q, dQ, cost, iters, status = result
    ndof, T = Q.shape[1], Q.shape[2]
    shards = _shard_indices(B, world, assignment)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    bad = torch.tensor([0 if np.array_equal(np.asarray(mine), shards[rank]) else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if int(bad.item()):
        raise ValueError("gather_results: on some rank `mine` is not that rank's shard under the given assignment")
    mx = max(len(ix) for ix in shards)
    width = ndof * T + ndof * (T - 1) + 1
    payload, ipay = np.zeros((mx, width)), np.zeros((mx, 2), dtype=np.int32)
    n = len(mine)
    payload[:n, : ndof * T] = np.asarray(Q).reshape(n, ndof * T)  # (explicit widths: a rank may hold no instance at all)
    payload[:n, ndof * T: ndof * T + ndof * (T - 1)] = np.asarray(dQ).reshape(n, ndof * (T - 1))
    payload[:n, -1] = cost
    ipay[:n, 0], ipay[:n, 1] = iters, status
    t0 = time.perf_counter()
    send, isend = torch.from_numpy(payload).to(dev), torch.from_numpy(ipay).to(dev)
    recv, irecv = [torch.empty_like(send) for _ in range(world)], [torch.empty_like(isend) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    dist.all_gather(irecv, isend, group=group)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    if stats is not None:
        stats.update(backend=backend, seconds=time.perf_counter() - t0, bytes_sent=int(payload.nbytes + ipay.nbytes),
                     bytes_received=int(world * (payload.nbytes + ipay.nbytes)), collectives=2)
    Qa, dQa = np.empty((B, ndof, T)), np.empty((B, ndof, T - 1))
    ca, ia, sa = np.empty(B), np.empty(B, np.int32), np.empty(B, np.int32)
    for r, idx in enumerate(shards):
        blk, iblk = recv[r].cpu().numpy()[: len(idx)], irecv[r].cpu().numpy()[: len(idx)]
        Qa[idx] = blk[:, : ndof * T].reshape(-1, ndof, T)
        dQa[idx] = blk[:, ndof * T: ndof * T + ndof * (T - 1)].reshape(-1, ndof, T - 1)
        ca[idx], ia[idx], sa[idx] = blk[:, -1], iblk[:, 0], iblk[:, 1]
    return np.arange(B), Qa, dQa, ca, ia, sa


def solve_local_shard(solve_fn: Callable[..., tuple], mine, scene_id, qc, goals, n_goals, standoff, base_pos, Q0,
                      B: int, rank: int, world: int, group=None, gather: bool = True,
                      assignment: Optional[np.ndarray] = None, stats: Optional[dict] = None):
    """Like solve_sharded, but the argument arrays hold ONLY this rank's instances (rows in the order of `mine`, the
    global indices of the shard): a rank never materialises the inputs of instances it does not solve.  B is the size of
    the whole batch."""
    mine = np.asarray(mine)
    n = len(mine)
    scene_id = np.broadcast_to(np.asarray(scene_id, dtype=np.int32), (n,))
    Q0 = np.asarray(Q0, dtype=np.float64)
    ndof, T = Q0.shape[-2], Q0.shape[-1]
    if n:
        result = solve_fn(scene_id, np.asarray(qc, dtype=np.float64).reshape(n, -1), np.asarray(goals, dtype=np.float64).reshape(n, -1, 16),
                          n_goals, standoff, base_pos, Q0.reshape(n, ndof, T))
    else:
        result = (np.empty((0, ndof, T)), np.empty((0, ndof, T - 1)), np.empty(0), np.empty(0, np.int32), np.empty(0, np.int32))
    if not gather or not _collective_applies(world, group):
        return (mine,) + tuple(result)
    return gather_results(mine, result, B, rank, world, group, assignment, stats)


def solve_sharded(solve_fn: Callable[..., tuple], scene_id, qc, goals, n_goals, standoff, base_pos, Q0,
                  rank: int, world: int, group=None, gather: bool = True, assignment: Optional[np.ndarray] = None):
    """Solve this rank's shard with `solve_fn` (SolverHandle.solve_batch signature) and, if `gather`,
    all_gather the results so every rank returns the full batch in the original order.

    The argument arrays cover the WHOLE batch (every rank picks its rows); solve_local_shard takes a rank's rows only.
    The caller is responsible for having uploaded the scenes its shard refers to on this rank.
    """
    scene_id = np.asarray(scene_id, dtype=np.int32)
    B = scene_id.shape[0]
    qc = np.asarray(qc, dtype=np.float64).reshape(B, -1)
    goals = np.asarray(goals, dtype=np.float64).reshape(B, -1, 16)
    n_goals = np.broadcast_to(np.asarray(n_goals, dtype=np.int32), (B,))
    base_pos = np.broadcast_to(np.asarray(base_pos, dtype=np.float64).reshape(-1, 3), (B, 3))
    Q0 = np.asarray(Q0, dtype=np.float64)
    so = None if standoff is None else np.broadcast_to(np.asarray(standoff, dtype=np.float64).reshape(-1, 16), (B, 16))
    mine = _shard_indices(B, world, assignment)[rank]
    return solve_local_shard(solve_fn, mine, scene_id[mine], qc[mine], goals[mine], n_goals[mine],
                             None if so is None else so[mine], base_pos[mine], Q0[mine], B, rank, world, group, gather, assignment)


_WQ_CALLS = [0]


def _default_store():
    """The key-value store of the default process group (TCPStore behind torch.distributed.run / init_process_group):
    its add() is an atomic fetch-and-add served by rank 0's store daemon, for RCCL and gloo groups alike."""
    import torch.distributed as dist
    return dist.distributed_c10d._get_default_store()


def scene_chunks(scene_id: Sequence[int], max_instances: int = 0) -> list:
    """The units of the work queue: the instances of one scene (a scene's cost field is then uploaded by exactly one rank,
    SURVEY.md 8e), split into runs of at most `max_instances` when that is given.  Chunks in the order of first appearance
    of their scene; indices ascending inside a chunk."""
    scene_id = np.asarray(scene_id)
    order, seen = [], {}
    for i, s in enumerate(scene_id.tolist()):
        if s not in seen:
            seen[s] = len(order)
            order.append([])
        order[seen[s]].append(i)
    out = []
    for idx in order:
        step = max_instances if max_instances > 0 else len(idx)
        out += [np.asarray(idx[i:i + step], dtype=np.int64) for i in range(0, len(idx), step)]
    return out


def solve_work_queue(solve_fns, chunks: Sequence[np.ndarray], make_args: Callable[[np.ndarray], tuple], B: int, rank: int, world: int,
                     group=None, store=None, gather: bool = True, key: Optional[str] = None, stats: Optional[dict] = None):
    """Dynamic balance across ranks ("a per-GPU work queue"): the chunks (scene_chunks) are claimed one at a
    time through ONE shared counter -- `store.add(key, 1)`, an atomic fetch-and-add on the process group's key-value store --
    so a rank that drew instances of 10 iterations takes the next chunk while a rank with 100-iteration instances is still
    busy; iteration counts of this workload span 10-100 and a static partition ends with its slowest rank.

    solve_fns: one callable (SolverHandle.solve_batch signature) or a list of them -- one claiming thread each (several
    handles of a GPU keep several chunks in flight, like the lanes of BatchPipeline); make_args(indices) -> the argument
    tuple (scene_id, qc, goals, n_goals, standoff, base_pos, Q0) of those instances, called by the claiming thread (it is
    where a rank uploads the scene of a chunk it drew).  Every instance is solved exactly once by somebody; results are
    the static partition's bit for bit (instances are independent), whatever the order.  With `gather` every rank returns
    the whole batch in the original order (two padded all_gathers as in gather_results, preceded by an all_gather of the
    index lists); without, its own instances (indices ascending).  `stats` receives chunks / instances claimed here."""
    import threading
    fns = list(solve_fns) if isinstance(solve_fns, (list, tuple)) else [solve_fns]
    n_chunks = len(chunks)
    counter = None
    if world > 1 or _group_exists():
        store = store if store is not None else _default_store()
        _WQ_CALLS[0] += 1  # every rank makes the same calls in the same order: the same key everywhere
        counter = f"gto_wq/{key if key is not None else _WQ_CALLS[0]}"
    local_next = [0]
    lock = threading.Lock()
    got: list = []
    errors: list = []

    def claim() -> int:
        if counter is not None:
            return int(store.add(counter, 1)) - 1
        with lock:
            local_next[0] += 1
            return local_next[0] - 1

    def worker(fn):
        try:
            while not errors:
                c = claim()
                if c >= n_chunks:
                    return
                idx = np.asarray(chunks[c])
                res = fn(*make_args(idx))
                with lock:
                    got.append((c, idx, res))
        except BaseException as e:  # noqa: BLE001 -- re-raised below, after the other threads have stopped claiming
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(fn,), name=f"gto-wq-{i}") for i, fn in enumerate(fns[1:], 1)]
    for t in threads:
        t.start()
    worker(fns[0])
    for t in threads:
        t.join()
    err_local = 1 if errors else 0
    if world > 1 or (gather and _collective_applies(world, group)):
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        flag = torch.tensor([err_local], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)  # nobody is left waiting in the gather
        if int(flag.item()):
            raise RuntimeError("solve_work_queue: a rank failed") from (errors[0] if errors else None)
    elif errors:
        raise errors[0]
    got.sort(key=lambda g: g[0])
    if got:
        mine = np.concatenate([g[1] for g in got])
        res = tuple(np.concatenate([np.asarray(g[2][i]) for g in got]) for i in range(5))
        order = np.argsort(mine, kind="stable")
        mine, res = mine[order], tuple(a[order] for a in res)
    else:
        a0 = make_args(np.zeros(0, dtype=np.int64))
        Q0 = np.asarray(a0[6], dtype=np.float64)
        ndof, T = Q0.shape[-2], Q0.shape[-1]
        mine = np.zeros(0, dtype=np.int64)
        res = (np.empty((0, ndof, T)), np.empty((0, ndof, T - 1)), np.empty(0), np.empty(0, np.int32), np.empty(0, np.int32))
    if stats is not None:
        stats.update(chunks_total=n_chunks, chunks_claimed=len(got), instances_claimed=int(len(mine)))
    if not gather or not _collective_applies(world, group):
        return (mine,) + res
    import torch.distributed as dist
    lists = [None] * world
    dist.all_gather_object(lists, mine.tolist(), group=group)
    assignment = np.full(B, -1, dtype=np.int32)
    for r, l_ in enumerate(lists):
        assignment[np.asarray(l_, dtype=np.int64)] = r
    if (assignment < 0).any():
        raise RuntimeError("solve_work_queue: instances nobody claimed (chunks do not cover the batch)")
    if stats is not None:
        stats.update(instances_by_rank=[len(l_) for l_ in lists])
    return gather_results(mine, res, B, rank, world, group, assignment, stats)


def merge_batches(batches: Sequence[tuple]):
    """Concatenate solve_batch argument tuples (scene_id, qc, goals, n_goals, standoff, base_pos, Q0) along the
    instance axis.  Scalars / single rows are broadcast per batch first; goal sets are padded to the widest
    n_max (n_goals keeps them ragged); standoff must be given for all batches or for none.
    Returns (merged tuple, instances per batch)."""
    if not batches:
        raise ValueError("nothing to merge")
    ndof = np.asarray(batches[0][6]).shape[-2]
    sizes = [np.asarray(b[1], dtype=np.float64).reshape(-1, ndof).shape[0] for b in batches]
    with_so = [b[4] is not None for b in batches]
    if any(with_so) and not all(with_so):
        raise ValueError("cannot merge batches with and without a standoff pose")
    n_max = max(np.asarray(b[2], dtype=np.float64).reshape(B, -1, 16).shape[1] for b, B in zip(batches, sizes))
    sid, qc, goals, ng, so, base, Q0 = [], [], [], [], [], [], []
    for b, B in zip(batches, sizes):
        sid.append(np.broadcast_to(np.asarray(b[0], dtype=np.int32), (B,)))
        qc.append(np.asarray(b[1], dtype=np.float64).reshape(B, ndof))
        g = np.asarray(b[2], dtype=np.float64).reshape(B, -1, 16)
        pad = np.zeros((B, n_max, 16))
        pad[:, :g.shape[1]] = g
        goals.append(pad)
        ng.append(np.broadcast_to(np.asarray(b[3], dtype=np.int32), (B,)))
        if b[4] is not None:
            so.append(np.broadcast_to(np.asarray(b[4], dtype=np.float64).reshape(-1, 16), (B, 16)))
        base.append(np.broadcast_to(np.asarray(b[5], dtype=np.float64).reshape(-1, 3), (B, 3)))
        Q0.append(np.asarray(b[6], dtype=np.float64).reshape(B, ndof, -1))
    cat = np.concatenate
    return (cat(sid), cat(qc), cat(goals), cat(ng), cat(so) if so else None, cat(base), cat(Q0)), sizes


def split_results(result: tuple, sizes: Sequence[int]) -> list:
    """Undo merge_batches on a solve_batch result tuple: one tuple per original batch."""
    edges = np.cumsum([0] + list(sizes))
    return [tuple(a[lo:hi] for a in result) for lo, hi in zip(edges[:-1], edges[1:])]


class BatchPipeline:
    """Several batches in flight on ONE GPU.

    A batch's solve is a chain of ~2*iters dependent kernel launches whose tail rounds only carry the
    few instances that are still iterating, so one batch of 64 leaves most of the 256 CUs idle
    (measured: B=64 19.7k traj/s, B=256 48k traj/s).  Instances of different
    batches are independent (gto/gto_planner.py:185-245 shares nothing across calls), so the pipeline
    keeps `depth` solver handles, each with its own HIP stream and scratch state, and lets `depth`
    host threads drive them concurrently; ctypes releases the GIL for the duration of each C call.
    Results are identical to solving the batches one after the other.

    `solvers` are objects of the SolverHandle surface that already hold the scenes they need.
    """

    def __init__(self, solvers: Sequence):
        import queue
        from concurrent.futures import ThreadPoolExecutor
        if len(solvers) < 1:
            raise ValueError("BatchPipeline needs at least one solver handle")
        self.depth = len(solvers)
        self._idle = queue.SimpleQueue()
        for s in solvers:
            self._idle.put(s)
        self._pool = ThreadPoolExecutor(max_workers=self.depth, thread_name_prefix="gto-pipe")

    def _run(self, method: str, args, kwargs):
        s = self._idle.get()
        try:
            return getattr(s, method)(*args, **kwargs)
        finally:
            self._idle.put(s)

    def submit(self, method: str, *args, **kwargs):
        """Queue `solver.<method>(*args, **kwargs)` on the next idle handle; returns a Future."""
        return self._pool.submit(self._run, method, args, kwargs)

    def solve_batches(self, batches: Sequence[tuple], merge: int = 1) -> list:
        """solve_batch over a list of argument tuples; results in submission order.

        merge > 1 folds that many consecutive batches into ONE solve_batch call (merge_batches) and splits the
        results again: a lane's launches are latency-bound while a batch of 64 fills a fraction of the GPU, so
        four batches per launch on each of four lanes nearly doubles the rate.  Instances
        are independent, so every batch gets exactly the results it would get alone."""
        if merge <= 1:
            futs = [self.submit("solve_batch", *b) for b in batches]
            return [f.result() for f in futs]
        groups = [list(batches[i:i + merge]) for i in range(0, len(batches), merge)]
        futs = [self.submit("solve_batch", *merge_batches(g)[0]) for g in groups]
        out = []
        for g, f in zip(groups, futs):
            out.extend(split_results(f.result(), [np.asarray(b[1]).reshape(-1, np.asarray(b[6]).shape[-2]).shape[0] for b in g]))
        return out

    def close(self):
        self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
