"""Multi-GPU layout of the front-end: camera streams are independent units (every stream owns its last pyramid, its
points and its feature database, like the per-camera state at src/track/TrackBase.h:350-359), so a job is sharded
by stream with one process per GPU and nothing is exchanged in the per-frame path.  The only communication is the
reduction of the run totals after the timed region."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def streams_of_rank(n_streams_total: int, world: int, rank: int) -> List[int]:
    """Round-robin assignment of global stream ids to ranks: stream s runs on rank s % world."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return list(range(rank, n_streams_total, world))


def weak_scaling_stream_ids(streams_per_rank: int, rank: int) -> List[int]:
    """Weak scaling: every rank runs the same number of streams; ids are disjoint across ranks so that the synthetic
    sequences (seeded by id) differ."""
    return [rank * 1000 + s for s in range(streams_per_rank)]


def reduce_run_totals(elapsed_s: float, counts: Sequence[float], device=None) -> Tuple[float, List[float]]:
    """Whole-job totals: the slowest rank's elapsed time and the sum of every counter over ranks.  Without an
    initialised process group the inputs are returned unchanged (single process)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), [float(c) for c in counts]
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor(list(counts), dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in c.tolist()]
