"""Rank program of tests/test_sharding_gloo.py: the synthesize command with the CPU oracle as the
model, started once per rank by torch.distributed.run (gloo, host only)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch  # noqa: E402

from oracle_shim import oracle_synthesizer  # noqa: E402
from tacotron_amd import cli, sharding  # noqa: E402


def main():
  torch.set_num_threads(1)
  mode = sys.argv[1]
  if mode == "synthesize":
    ns = cli.build_parser().parse_args(sys.argv[1:])
    slow_rank = os.environ.get("SHARD_WORKER_SLOW_RANK")

    def factory(*args, **kwargs):
      # one rank takes long to build its model: the others have written files by the time it goes on
      if slow_rank is not None and int(os.environ.get("RANK", "0")) == int(slow_rank):
        import time
        time.sleep(3.0)
      return oracle_synthesizer(*args, **kwargs)

    if os.environ.get("SHARD_WORKER_FAIL_PLAN") and int(os.environ.get("RANK", "0")) == 0:
      def broken_plan(*args, **kwargs):   # rank 0 cannot plan (an unreadable output folder, say)
        raise PermissionError("output folder cannot be listed")
      cli.plan_outputs = broken_plan
    ok = ns.handler(ns, synthesizer_factory=factory)
    rank, n = sharding.world()
    print(f"rank {rank}/{n} ok={ok}", flush=True)
    sys.exit(0 if ok else 1)
  if mode == "fail":                     # rank 1's work raises: rank 0 must see it, nobody may hang
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, n = sharding.world()

    def work(indices):
      if rank == 1:
        raise ValueError("boom on rank 1")
      return {i: i * i for i in indices}

    try:
      sharding.run_sharded(6, [1.0] * 6, work)
    except Exception as ex:  # pylint: disable=broad-except
      print(f"rank {rank} raised {type(ex).__name__}: {ex}", flush=True)
      dist.destroy_process_group()
      sys.exit(0)
    print(f"rank {rank} did not raise", flush=True)
    sys.exit(3)
  raise SystemExit(f"unknown mode {mode}")


if __name__ == "__main__":
  main()
