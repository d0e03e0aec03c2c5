"""The N > 1 path: lines partitioned over one process per GPU, host-side gather of the records.
Run here with world_size 2 over gloo and the CPU oracle in place of the GPU model."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from tacotron_amd import sharding

TINY_CKPT = os.path.join(GOLDEN, "tiny_checkpoint.pt")
WORKER = os.path.join(ROOT, "tests", "shard_worker.py")


def test_lpt_partition_covers_balances_and_is_deterministic():
  r = np.random.RandomState(0)
  costs = r.randint(20, 201, size=1000).astype(float)
  for n in (1, 2, 4, 8):
    shards = sharding.lpt_partition(costs, n)
    assert sorted(i for s in shards for i in s) == list(range(1000))
    loads = sharding.shard_loads(costs, shards)
    assert max(loads) - min(loads) <= costs.max()              # LPT bound for equal-speed shards
    assert shards == sharding.lpt_partition(list(costs), n)
    for s in shards:                                           # descending cost inside a shard
      assert all(costs[a] >= costs[b] for a, b in zip(s, s[1:]))
  assert sharding.lpt_partition([], 3) == [[], [], []]
  assert sharding.lpt_partition([5.0], 2) == [[0], []]
  with pytest.raises(ValueError):
    sharding.lpt_partition([1.0], 0)
  assert sharding.batches([4, 2, 9, 1, 7], 2) == [[4, 2], [9, 1], [7]]


def test_run_sharded_single_process_orders_records():
  out = sharding.run_sharded(5, [3, 1, 4, 1, 5], lambda idx: {i: f"r{i}" for i in idx})
  assert out == ["r0", "r1", "r2", "r3", "r4"]
  with pytest.raises(RuntimeError):
    sharding.run_sharded(3, [1, 1, 1], lambda idx: {})         # a record is missing


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _launch(n, args, timeout=300, extra_env=None):
  env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT, **(extra_env or {}))
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), WORKER] + args
  return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_write_the_same_files_as_one(tmp_path):
  r = np.random.RandomState(4)
  lines = []
  for i in range(11):
    if i in (4, 8):
      lines.append("")
    lines.append("|".join(f"s{r.randint(1, 31)}" for _ in range(r.randint(2, 12))))
  text = tmp_path / "in.txt"
  text.write_text("\n".join(lines), encoding="UTF-8")
  common = ["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-seed", "5", "--max-decoder-steps", "14"]
  one = _launch(1, common + ["-out", str(tmp_path / "one")])
  assert one.returncode == 0, one.stdout + one.stderr
  two = _launch(2, common + ["-out", str(tmp_path / "two"), "--batch-size", "2"])
  assert two.returncode == 0, two.stdout + two.stderr
  assert "rank 0/2 ok=True" in two.stdout and "rank 1/2 ok=True" in two.stdout
  names = sorted(p.name for p in (tmp_path / "one").iterdir())
  assert len(names) == 11 and names == sorted(p.name for p in (tmp_path / "two").iterdir())
  for n in names:
    np.testing.assert_array_equal(np.load(tmp_path / "one" / n), np.load(tmp_path / "two" / n))


def test_a_slow_rank_plans_the_same_lines_without_overwrite(tmp_path):
  """Without --overwrite a line whose output exists is skipped and does not advance the utterance counter.  A rank
  that builds its model slowly must not take the files the fast rank has already written for existing outputs:
  the plan is made once, before anybody writes."""
  r = np.random.RandomState(9)
  lines = ["|".join(f"s{r.randint(1, 31)}" for _ in range(r.randint(2, 9))) for _ in range(8)]
  text = tmp_path / "in.txt"
  text.write_text("\n".join(lines), encoding="UTF-8")
  common = ["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-seed", "5", "--max-decoder-steps", "10"]
  one = _launch(1, common + ["-out", str(tmp_path / "one")])
  assert one.returncode == 0, one.stdout + one.stderr
  two = _launch(2, common + ["-out", str(tmp_path / "two")], extra_env={"SHARD_WORKER_SLOW_RANK": "1"})
  assert two.returncode == 0, two.stdout + two.stderr
  names = sorted(p.name for p in (tmp_path / "one").iterdir())
  assert len(names) == 8 and names == sorted(p.name for p in (tmp_path / "two").iterdir())
  for n in names:
    np.testing.assert_array_equal(np.load(tmp_path / "one" / n), np.load(tmp_path / "two" / n))


def test_a_failed_plan_on_rank_0_ends_every_rank(tmp_path):
  """Only rank 0 plans the output files; when that fails, the other ranks (waiting for the plan) must get the error
  and end with a failure status instead of waiting for the process group's timeout."""
  text = tmp_path / "in.txt"
  text.write_text("s1|s2|s3\ns4|s5", encoding="UTF-8")
  res = _launch(2, ["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-seed", "5", "--max-decoder-steps", "6",
                    "-out", str(tmp_path / "out")], timeout=120, extra_env={"SHARD_WORKER_FAIL_PLAN": "1"})
  assert res.returncode != 0
  assert "rank 0/2 ok=False" in res.stdout and "rank 1/2 ok=False" in res.stdout, res.stdout + res.stderr
  assert "output folder cannot be listed" in res.stdout + res.stderr
  assert not (tmp_path / "out").exists()


def test_a_failing_rank_is_reported_on_every_rank_without_a_hang():
  res = _launch(2, ["fail"], timeout=120)
  assert res.returncode == 0, res.stdout + res.stderr
  assert "rank 0 raised RuntimeError: rank 1 failed" in res.stdout
  assert "rank 1 raised ValueError: boom on rank 1" in res.stdout


def test_bench_rank_plumbing_rehearsal():
  """bench.py's multi-process skeleton (launch by torch.distributed.run, barriers, max over ranks, one
  JSON line from rank 0) with sleeps in place of GPU work: the slowest rank must set the step time."""
  import json
  env = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
         "--rehearse-ranks"]
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
  assert res.returncode == 0, res.stdout + res.stderr
  lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1
  out = json.loads(lines[0])
  assert out["n_gpus"] == 2 and out["steps"] == 5 and out["rehearsal"] is True
  # the sharded line set of the multi-GPU workload: every line in exactly one rank's shard, shards of equal size here
  assert out["lines"] == 512 and out["lines_covered_once"] is True and 200 <= out["rank0_lines"] <= 312
  assert out["ms_per_step"] >= 4.0          # rank 1 sleeps 4 ms per step, rank 0 only 2 ms
  # the one-GPU figure of the same workload and the efficiency against it come out of the same run
  assert out["one_gpu_same_workload"]["ms_per_step"] >= 2.0
  assert 0.3 < out["efficiency"] < 0.7      # 2 ranks, the slower at half speed: about 0.5
  assert out["cpu_baseline"] is not None


def test_bench_started_plainly_spawns_its_ranks():
  """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (how a driver may start it): the process starts
  the two ranks itself as children, before anything touches a GPU, and relays rank 0's one JSON line."""
  import json
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
  env.update(OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--rehearse-ranks"]
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
  assert res.returncode == 0, res.stdout + res.stderr
  lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1
  out = json.loads(lines[0])
  assert out["n_gpus"] == 2 and out["steps"] == 4 and out["rehearsal"] is True
  assert out["lines_covered_once"] is True and "efficiency" in out
  # a wrong launch is still refused: --gpus must agree with the launcher's world size
  bad = subprocess.run(cmd, env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=60)
  assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stdout + bad.stderr)
