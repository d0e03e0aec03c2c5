"""Sharding of independent utterances over one process per GPU.

Utterances share no state (the reference synthesizes them one after another with the same seed:
src/tacotron_cli/inference.py:137-141, 236-242), so the N-GPU path is a partition of the line list
and a host-side gather of small per-line records.  There is no data-path collective: each rank
holds a full weight replica, synthesizes its own lines and writes its own output files.

The partition is longest-processing-time-first on a predicted cost per line (decoder frames grow
with the number of input symbols), which keeps the ranks' finishing times close when line lengths
vary widely.
"""
from __future__ import annotations

import heapq
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple


def predicted_cost(n_symbols: int) -> float:
  """Relative cost of one line: decoder frames are roughly proportional to the symbol count, the
  encoder and the per-utterance launch add a constant."""
  return float(n_symbols) + 8.0


def lpt_partition(costs: Sequence[float], n_shards: int) -> List[List[int]]:
  """Longest-processing-time-first: items in descending cost (ties: lower index first) each go to
  the currently lightest shard (ties: lower shard first).  Deterministic, so every rank computes
  the same partition without communicating.  Each shard lists its items in assignment order,
  i.e. by descending cost, which also puts lines of similar length next to each other for
  batching."""
  if n_shards < 1:
    raise ValueError("n_shards must be >= 1")
  order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
  heap: List[Tuple[float, int]] = [(0.0, s) for s in range(n_shards)]
  shards: List[List[int]] = [[] for _ in range(n_shards)]
  for i in order:
    load, s = heapq.heappop(heap)
    shards[s].append(i)
    heapq.heappush(heap, (load + float(costs[i]), s))
  return shards


def shard_loads(costs: Sequence[float], shards: Sequence[Sequence[int]]) -> List[float]:
  return [sum(float(costs[i]) for i in shard) for shard in shards]


def batches(shard: Sequence[int], batch_size: int) -> List[List[int]]:
  """Consecutive groups of at most ``batch_size`` items of a shard (already length-sorted)."""
  if batch_size < 1:
    raise ValueError("batch_size must be >= 1")
  return [list(shard[i:i + batch_size]) for i in range(0, len(shard), batch_size)]


def world() -> Tuple[int, int]:
  """(rank, world_size) of the initialised process group, or (0, 1) without one."""
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(), dist.get_world_size()
  return 0, 1


def run_sharded(n_items: int, costs: Sequence[float], work: Callable[[List[int]], Dict[int, Any]],
                group=None) -> Optional[List[Any]]:
  """Every rank calls this with the same ``n_items`` / ``costs``.  ``work(indices)`` processes this
  rank's items and returns ``{index: record}``; the records (small, picklable, host memory) are
  gathered on rank 0, which gets the list ordered by item index; other ranks get ``None``.
  A rank whose ``work`` raises reports the exception instead of records, and rank 0 re-raises it
  after the gather so that no rank is left waiting."""
  import torch.distributed as dist
  rank, n = world()
  if len(costs) != n_items:
    raise ValueError("one cost per item is required")
  shard = lpt_partition(costs, n)[rank]
  try:
    local: Any = work(list(shard))
    missing = set(shard) - set(local.keys())
    if missing:
      raise RuntimeError(f"rank {rank} returned no record for items {sorted(missing)[:8]}")
  except Exception as ex:  # carried to rank 0 below
    local = ex
  if n == 1:
    if isinstance(local, Exception):
      raise local
    return [local[i] for i in range(n_items)]
  gathered: Optional[List[Any]] = [None] * n if rank == 0 else None
  dist.gather_object(local, gathered, dst=0, group=group)
  failed = isinstance(local, Exception)
  if rank != 0:
    if failed:
      raise local
    return None
  merged: Dict[int, Any] = {}
  for r, part in enumerate(gathered):
    if isinstance(part, Exception):
      raise RuntimeError(f"rank {r} failed: {part!r}") from part
    merged.update(part)
  if len(merged) != n_items:
    raise RuntimeError(f"gather returned {len(merged)} of {n_items} records")
  return [merged[i] for i in range(n_items)]
