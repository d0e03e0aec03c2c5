#!/usr/bin/env python3
"""Benchmark of the hot path: Tacotron 2 mel inference at batch 1 on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one whole utterance through the C ABI: input vectors + encoder + persistent decoder
(``--frames`` decoder frames) + postnet, on the LJS-sized synthetic checkpoint (default HParams:
E=517, M=534, 28.5 M parameters) and the 102-symbol "North Wind" line of the reference's README.
Inputs (ids, prenet keep-masks) are resident in HBM before the timed region.  The synthetic
weights have no meaningful stop token, so the stop threshold is put out of reach and every
utterance runs the full ``--frames`` decoder steps (the reference's max_decoder_steps path).

With N > 1 (one process per GPU, launched by torch.distributed.run) every rank synthesizes its own
utterances: utterances are independent, there is no data-path collective; ranks only meet at the
timing barriers.  ``value`` is the whole-job mel frames per second.

Prints ONE JSON line (rank 0).  ``roofline`` prices the persistent decoder kernel: algorithmic
bytes per frame (weights touched + attention memory + recurrent state) x frames per launch over
the kernel's duration, measured with HIP events on the launch stream.  ``cpu_baseline`` times the
CPU oracle (a port of the reference's arithmetic, oracle/taco_oracle.c) on this host's cores on
a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak


def build_inputs(ckpt):
  """Ids of the North-Wind line through this package's own front-end (no GPU needed)."""
  from tacotron_amd import frontend, scenarios
  from tacotron_amd.checkpoint import get_hparams
  hp = get_hparams(ckpt)
  cores, stresses, _, _ = frontend.get_map_keys(scenarios.NORTH_WIND_SYMBOLS, hp)
  sym = np.array([[ckpt["symbol_mapping"][c] for c in cores]], dtype=np.int64)
  st = np.array([[ckpt["stress_mapping"][s] for s in stresses]], dtype=np.int64)
  spk = np.full_like(sym, ckpt["speaker_mapping"]["Linda Johnson"])
  return sym, st, spk


def cpu_baseline(ckpt, sym, st, spk, target_s):
  """The oracle (CPU port of the reference's arithmetic) on this host's cores, bounded sample."""
  from oracle.oracle import OracleModel, build, threads
  from tacotron_amd.masks import prenet_keep_masks
  build()
  om = OracleModel(ckpt)
  never = 2.0
  probe = 60
  masks = prenet_keep_masks(1111, probe, om.dims.P).numpy()
  t0 = time.perf_counter()
  mem, pm = om.encode([sym, st], spk, None)
  t_enc = time.perf_counter() - t0
  t0 = time.perf_counter()
  om.decode(mem, pm, None, masks, probe, gate_threshold=never, want_align=True)
  per_frame = (time.perf_counter() - t0) / probe
  frames = int(min(60000, max(200, target_s / max(per_frame, 1e-6))))
  masks = prenet_keep_masks(1111, frames, om.dims.P).numpy()
  t0 = time.perf_counter()
  mem, pm = om.encode([sym, st], spk, None)
  mel, gate, align, n_frames, reached, _ = om.decode(mem, pm, None, masks, frames, gate_threshold=never)
  om.postnet(mel, n_frames)
  dt = time.perf_counter() - t0
  return dict(value=round(frames / dt, 2), unit="mel frames/s", cores=threads(), kind="port",
              sample=f"1 utterance, T_in=102, {frames} decoder frames + encoder + postnet, {dt:.1f} s wall "
                     f"(encoder alone {t_enc * 1e3:.0f} ms)",
              ms_per_frame=round(dt / frames * 1e3, 4))


def timed_region(steps, step, barrier, max_over_ranks):
  """EXACTLY `steps` steps between two barriers; returns the slowest rank's seconds."""
  barrier()
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  barrier()
  return max_over_ranks(time.perf_counter() - t0)


def rehearse(args, world, rank):
  """The launch / barrier / reduction skeleton of main() with a sleep in place of the GPU step."""
  dist = None
  if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo")

  def barrier():
    if dist is not None:
      dist.barrier()

  def max_over_ranks(seconds):
    if dist is None:
      return seconds
    t = torch.tensor([seconds], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

  step_s = 0.002 * (1 + rank)                    # ranks of unequal speed: the slowest must set the time
  for _ in range(args.warmup):
    time.sleep(step_s)
  elapsed = timed_region(args.steps, lambda: time.sleep(step_s), barrier, max_over_ranks)
  if rank == 0:
    print(json.dumps({"rehearsal": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                      "value": round(world * args.steps * args.frames / elapsed, 1)}), flush=True)
  if dist is not None:
    dist.destroy_process_group()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--frames", type=int, default=1000, help="decoder frames per utterance (max_decoder_steps)")
  ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample size; 0 disables it")
  ap.add_argument("--rehearse-ranks", action="store_true",
                  help="no GPU work: every rank sleeps instead of synthesizing and the ranks meet over gloo; "
                       "exercises the multi-process launch, barrier and max-over-ranks plumbing on a CPU-only machine "
                       "(the printed line is marked rehearsal and is not a measurement)")
  args = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                       "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
  if args.rehearse_ranks:
    return rehearse(args, world, rank)
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X; there is no CPU path (the cpu_baseline leg is only a side measurement)")
  torch.cuda.set_device(local_rank)
  device = torch.device("cuda", local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=device)

  from tacotron_amd import abi, scenarios
  from tacotron_amd.checkpoint import get_hparams
  from tacotron_amd.masks import prenet_keep_masks
  from tacotron_amd.model import Tacotron2

  ckpt = scenarios.checkpoint("ljs")
  hp = get_hparams(ckpt)
  model = Tacotron2(hp, **abi.mapping_counts(ckpt, hp))
  model.load_state_dict(ckpt["model_state"])
  model.to(device)
  sym, st, spk = build_inputs(ckpt)
  sym_d, st_d, spk_d = (torch.from_numpy(a).to(device) for a in (sym, st, spk))
  frames = args.frames
  masks = prenet_keep_masks(1111, frames, model.dims.P).to(device)
  never = 2.0                                    # sigmoid(gate) never exceeds it

  def step():
    mem, pm = model.encode([sym_d, st_d], spk_d, None)
    mel, gate, align, n_frames, reached = model.decode(mem, pm, None, masks, frames, gate_threshold=never,
                                                       want_align=True)
    post = model.postnet(mel, n_frames)
    return post, n_frames

  def barrier():
    torch.cuda.synchronize(device)
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize(device)

  for _ in range(args.warmup):
    post, n_frames = step()
  assert int(n_frames[0]) == frames
  dec_us = []

  def timed_step():
    step()
    dec_us.append(model.last_decode_timing()[0])   # decode has completed already: the call returns at once

  def max_over_ranks(seconds):
    if dist is None:
      return seconds
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

  elapsed = timed_region(args.steps, timed_step, barrier, max_over_ranks)

  # per-frame step latency: one more utterance, outside the timed region, with the kernel's frame clock on
  model.record_frame_clock(frames)
  step()
  torch.cuda.synchronize(device)
  lat = np.sort(model.frame_latencies_us(frames).numpy()[8:])      # the first frames run before the naps have settled
  model.record_frame_clock(0)
  step_latency = {"p50": round(float(lat[len(lat) // 2]), 2), "p99": round(float(lat[int(len(lat) * 0.99)]), 2),
                  "max": round(float(lat[-1]), 2), "frames": int(len(lat))}

  total_frames = world * args.steps * frames
  value = total_frames / elapsed
  d = model.dims
  T_in = sym.shape[1]
  bytes_per_frame = model.step_weight_bytes + T_in * (d.A + d.M) * 4 + \
    (2 * d.R_a + 2 * d.R_d + d.M + d.P + d.n_mel + 1 + 2 * T_in) * 4
  dec_s = float(np.mean(dec_us)) * 1e-6
  achieved = bytes_per_frame * frames / dec_s / 1e9
  hop = hp.hop_length
  audio_s = (frames - 1) * hop / 22050.0
  # HBM bytes of the decoder kernel from the committed PMC profile (rocprofv3 cannot run inside this
  # process); per frame, scaled to this launch's frame count
  traffic = None
  try:
    import glob
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
    with open(latest) as f:
      traffic = int(json.load(f)["hbm_bytes_per_frame"] * frames)
  except (IndexError, OSError, KeyError, ValueError):
    traffic = None
  out = {
    "metric": "mel frames/sec", "value": round(value, 1), "unit": "mel frames/s", "n_gpus": world,
    "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
    "config": {"workload": f"LJS-sized synthetic checkpoint (E=517, M=534), North-Wind line T_in={T_in}, B=1, "
                           f"{frames} decoder frames per utterance (stop disabled), encoder + persistent decoder + postnet",
               "frames_per_step": frames, "utterances_per_step": 1, "parallelism": f"{world} independent replica(s)"},
    "real_time_factor": round(world * args.steps * audio_s / elapsed, 1),
    "decoder_us_per_frame": round(dec_s / frames * 1e6, 3),
    "decoder_step_latency_us": step_latency,
    "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                 "kernel": "taco_decoder_kernel", "bytes_per_frame": int(bytes_per_frame),
                 "kernel_ms_per_launch": round(dec_s * 1e3, 4),
                 "note": "achieved = algorithmic bytes (every weight once per frame) / time; the weights are "
                         "register-resident, so a fraction above 1 means faster than streaming them from HBM could be"},
  }
  if rank == 0 and world == 1 and args.cpu_seconds > 0:
    out["cpu_baseline"] = cpu_baseline(ckpt, sym, st, spk, args.cpu_seconds)
  elif rank == 0:
    out["cpu_baseline"] = None
  if rank == 0:
    print(json.dumps(out), flush=True)
  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
