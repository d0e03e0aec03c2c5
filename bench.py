#!/usr/bin/env python3
"""Benchmark of the hot path: Tacotron 2 mel inference at batch 1 on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one whole utterance through the C ABI: input vectors + encoder + persistent decoder
(``--frames`` decoder frames) + postnet, on the LJS-sized synthetic checkpoint (default HParams:
E=517, M=534, 28.5 M parameters) and the 102-symbol "North Wind" line of the reference's README.
Inputs (ids, prenet keep-masks) are resident in HBM before the timed region.  The synthetic
weights have no meaningful stop token, so the stop threshold is put out of reach and every
utterance runs the full ``--frames`` decoder steps (the reference's max_decoder_steps path).

With N > 1 (one process per GPU, launched by torch.distributed.run) the workload is the batch-synthesis
configuration: N x ``--lines-per-gpu`` synthetic lines (T_in ~ U{20..200}, generator seed 2024), every one run
to ``--frames`` decoder frames, partitioned over the ranks longest-processing-time-first on predicted cost
(tacotron_amd.sharding, the partition the ``synthesize`` command uses), each rank decoding its shard in padded
batches of 64 (one lockstep launch each, sixteen utterances in flight).  Utterances are independent: there is no data-path collective and
the ranks only meet at the timing barriers.  One step = one pass over the whole line set; ``value`` is the
whole-job mel frames per second.  ``--workload lines`` runs the same on one GPU.

Prints ONE JSON line (rank 0).  ``roofline`` prices the dominant kernel of the workload, timed with HIP events
on the launch stream.  Batch 1 (taco_decoder_kernel): its weights never leave the register file, so neither HBM
bandwidth nor the matrix cores bound a frame - the chain of five cross-CU hand-offs does; the record gives the
frame time against that floor (hops x the measured cost of one hop, scripts/hop_bench.hip), and beside it
``residency_ratio`` (algorithmic bytes per frame / time, over the HBM peak: above 1 = faster than streaming the
weights could be) and the LDS / issue figures of the committed counter profile.  Batches
(taco_decoder_batch_kernel): dense f32 contractions on the matrix cores - achieved TFLOP/s of the algorithmic
FLOPs against the 157.3 TFLOP/s f32 MFMA peak.  ``cpu_baseline`` times the
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


def spawn_ranks(n):
  """`python bench.py --gpus N` without a launcher: N ranks under torch.distributed.run as a child process (nothing in
  this process has touched the GPU), same arguments, rank 0's JSON line relayed on stdout."""
  import socket
  import subprocess
  with socket.socket() as sock:
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
  for line in child.stdout:
    sys.stdout.write(line)
    sys.stdout.flush()
  sys.exit(child.wait())


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

  # the lines workload's partition, as every rank computes it for itself: together the shards must cover the line set once
  from tacotron_amd import sharding
  n_lines = world * args.lines_per_gpu
  all_lens = np.random.RandomState(2024).randint(20, 201, size=n_lines).tolist()
  shards = sharding.lpt_partition([sharding.predicted_cost(n) for n in all_lens], world)
  mine = sorted(shards[rank])
  covered = None
  if dist is not None:
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    covered = sorted(i for part in parts for i in part) == list(range(n_lines))
  else:
    covered = mine == list(range(n_lines))
  step_s = 0.002 * (1 + rank)                    # ranks of unequal speed: the slowest must set the time
  for _ in range(args.warmup):
    time.sleep(step_s)
  elapsed = timed_region(args.steps, lambda: time.sleep(step_s), barrier, max_over_ranks)
  value = world * args.steps * args.frames / elapsed
  out = {"rehearsal": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
         "lines": n_lines, "lines_covered_once": covered, "rank0_lines": len(mine),
         "ms_per_step": round(elapsed / args.steps * 1e3, 4), "value": round(value, 1)}
  if world > 1:
    # the one-GPU figure of the same workload, measured in the same run on every rank (here: everybody sleeps rank 0's step)
    one_s = 0.002
    for _ in range(args.warmup):
      time.sleep(one_s)
    elapsed1 = timed_region(args.steps, lambda: time.sleep(one_s), barrier, max_over_ranks)
    one_gpu = args.steps * args.frames / elapsed1
    out["one_gpu_same_workload"] = {"value": round(one_gpu, 1), "ms_per_step": round(elapsed1 / args.steps * 1e3, 4)}
    out["efficiency"] = round(value / (world * one_gpu), 4)
  out["cpu_baseline"] = {"rehearsal": True, "rank": rank} if rank == 0 else None
  barrier()
  if rank == 0:
    print(json.dumps(out), flush=True)
  if dist is not None:
    dist.destroy_process_group()


MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input matrix-core peak (v_mfma_f32_16x16x4_f32 / 32x32x2_f32)
HOPS_PER_FRAME = 5             # dh -> h1 -> p -> ah -> energy partials -> dh (taco_decoder.hip)


def latest_profile(pattern):
  import glob
  found = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
  return found[-1] if found else None


def hop_cost():
  """The cost of one dependent cross-CU hand-off of a short vector (a small group of workgroups publishes, all 256
  consume, store to consumed), as the stand-alone microbenchmark measured it on this part: read from the newest
  profiles/r*_hop.json (written by scripts/micro_summary.py from the output of scripts/hop_probe.hip).  Returns
  (microseconds, file) or (None, None) when no such file is there."""
  src = latest_profile("r*_hop.json")
  if not src:
    return None, None
  try:
    with open(src) as f:
      return float(json.load(f)["hop_us"]), os.path.relpath(src, ROOT)
  except (OSError, KeyError, TypeError, ValueError):
    return None, None


def flops_per_frame(d, T_in):
  """Algorithmic FLOPs of one decoder frame of one utterance: 2 x the weights touched + the attention."""
  weights = d.P * d.n_mel + d.P * d.P + 4 * d.R_a * (d.P + d.M + d.R_a) + d.A * d.R_a + 4 * d.R_d * (d.R_a + d.M + d.R_d) + \
    (d.n_mel + 1) * (d.R_d + d.M)
  attention = T_in * (2 * d.K * d.A + 3 * d.A + d.M)      # location term, energies, context
  return 2 * (weights + attention)


MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix-core peak


def batch64_bf16_record(model, ckpt, device, fp32_record, frames=600, reps=3):
  """The same batch with the lockstep decoder's opt-in bf16 setting (LSTM gate rows and folded prenet rows held and
  multiplied as bf16, fp32 accumulation and cell states).  Not the headline: its frames are within 2e-3 of the
  reference's under teacher forcing (tests/test_gpu_parity.py), not within the fp32 bound."""
  model.set_precision("bf16")
  try:
    r = batch64_record(model, ckpt, device, frames, reps, pmc_pattern="r*_bf16_lockstep_pmc.json")
  finally:
    model.set_precision("fp32")
  roof = r["roofline"]
  return {"workload": r["workload"] + ", bf16 setting (taco_set_precision)", "value": r["value"], "unit": r["unit"],
          "ms_per_batch": r["ms_per_batch"], "decoder_us_per_lockstep_frame": r["decoder_us_per_lockstep_frame"],
          "decoder_us_per_utterance_frame": r["decoder_us_per_utterance_frame"],
          "speedup_over_fp32": round(r["value"] / fp32_record["value"], 3), "dtype": "bf16 products, f32 accumulation",
          "tolerance": "mel frames within 2e-3 of the reference's fp32 frames under teacher forcing (4.4e-4 measured over 1000 "
                       "frames); stop frames may differ from the fp32 ones",
          "roofline": {"bound": "mfma", "achieved": roof["achieved"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(roof["achieved"] / MFMA_BF16_PEAK_TFLOPS, 4), "kernel": "taco_decoder_batch_kernel<bf16>",
                       "kernel_ms_per_launch": roof["kernel_ms_per_launch"], "flops_per_launch": roof["flops_per_launch"],
                       "traffic": roof["traffic"], "traffic_source": roof["traffic_source"], "counters": roof["counters"],
                       "note": "same algorithmic FLOPs as the fp32 launch; in this setting the matrix cores are nowhere near "
                               "their bound - the frame is the chain of six dependent exchanges between the CUs"}}


def batch64_record(model, ckpt, device, frames=600, reps=3, pmc_pattern="r*[0-9a-z]_lockstep_pmc.json"):
  """The batched configuration: 64 synthetic lines (40..160 symbols, generator seed 2024, longest first) in one padded batch,
  every line run to `frames` decoder steps: encode + lockstep decoder (one persistent launch, 16 utterances in flight) + postnet."""
  from tacotron_amd import scenarios
  from tacotron_amd.masks import prenet_keep_masks
  lens = sorted(np.random.RandomState(2024).randint(40, 161, size=64).tolist(), reverse=True)
  ids = scenarios.random_ids(ckpt, lens, seed=2024)
  d = lambda a: None if a is None else torch.from_numpy(a).to(device)   # noqa: E731
  chans = [d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None]
  spk, ln = d(ids["speakers"]), d(ids["lens"])
  masks = prenet_keep_masks(1111, frames, model.dims.P).to(device)

  def run():
    mem, pm = model.encode(chans, spk, ln)
    mel, gate, align, nf, reached = model.decode(mem, pm, ln, masks, frames, gate_threshold=2.0, want_align=False)
    return model.postnet(mel, nf), nf

  run()
  torch.cuda.synchronize(device)
  dec_us = []
  t0 = time.perf_counter()
  for _ in range(reps):
    post, nf = run()
    dec_us.append(model.last_decode_timing()[0])
  torch.cuda.synchronize(device)
  dt = (time.perf_counter() - t0) / reps
  total = int(nf.sum())
  dec_s = float(np.mean(dec_us)) * 1e-6
  flops = sum(flops_per_frame(model.dims, t) for t in lens) * frames
  achieved = flops / dec_s / 1e12
  # bytes-based view of the same launch: the weights of a frame once per lockstep frame (16 utterances share them) plus every
  # utterance's memory, processed memory and state - what a kernel streaming its weights from HBM would move
  d_ = model.dims
  per_utt = [t * (d_.A + d_.M) * 4 + (2 * d_.R_a + 2 * d_.R_d + d_.M + d_.P + d_.n_mel + 1 + 2 * t) * 4 for t in lens]
  lockstep_frames = 64 * frames // 16
  alg_bytes = model.step_weight_bytes * lockstep_frames + sum(per_utt) * frames
  import glob
  # (the fp32 setting's counter files are r<tag>_lockstep_pmc.json, the bf16 setting's r<tag>_bf16_lockstep_pmc.json)
  found = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", pmc_pattern))
                 if ("_bf16_" in os.path.basename(f)) == ("_bf16_" in pmc_pattern))
  counters, traffic, src = None, None, (found[-1] if found else None)
  if src:
    prof = json.load(open(src))["derived"]
    counters = {"source": os.path.relpath(src, ROOT),
                "mfma_busy_share_of_simd_cycles": prof.get("mfma_busy_share_of_simd_cycles"),
                "mfma_instructions_per_lockstep_frame": prof.get("SQ_INSTS_VALU_MFMA_F32_per_lockstep_frame"),
                "issue_fraction_of_wave_cycles": prof.get("SQ_ACTIVE_INST_ANY_per_WAVE_CYCLE"),
                "valu_issue_fraction": prof.get("SQ_ACTIVE_INST_VALU_per_WAVE_CYCLE"),
                "scalar_issue_fraction": prof.get("SQ_ACTIVE_INST_SCA_per_WAVE_CYCLE"),
                "wait_fraction_of_wave_cycles": prof.get("SQ_WAIT_INST_ANY_per_WAVE_CYCLE"),
                "lds_bank_conflict_fraction": prof.get("lds_bank_conflict_cycles_per_active_cycle")}
    if prof.get("hbm_bytes_per_lockstep_frame"):
      traffic = int(prof["hbm_bytes_per_lockstep_frame"]) * (64 * frames // 16)
  return {"workload": f"64 lines of {min(lens)}..{max(lens)} symbols in one padded batch, {frames} frames each, lockstep decoder "
                      "(one persistent launch, 16 utterances in flight)",
          "value": round(total / dt, 1), "unit": "mel frames/s", "ms_per_batch": round(dt * 1e3, 3),
          "decoder_us_per_lockstep_frame": round(dec_s / (4 * frames) * 1e6, 3),
          "decoder_us_per_utterance_frame": round(dec_s / total * 1e6, 4),
          "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                       "traffic_source": counters and counters["source"], "counters": counters, "kernel": "taco_decoder_batch_kernel",
                       "flops_per_launch": int(flops), "kernel_ms_per_launch": round(dec_s * 1e3, 4),
                       "hbm_streaming": {"bytes_per_launch": int(alg_bytes), "achieved_gbs": round(alg_bytes / dec_s / 1e9, 1),
                                         "peak_gbs": HBM_PEAK_GBS, "frac": round(alg_bytes / dec_s / 1e9 / HBM_PEAK_GBS, 4)},
                       "note": "algorithmic FLOPs (2 x weights touched + attention, per utterance and frame) of the one launch that "
                               "decodes the batch of 64 (16 utterances in flight, slots refilled) / its device time (HIP events on "
                               "the launch stream); exact-f32 matrix-core products run at the f32 vector rate, 1/16 of the bf16 rate"}}


def lines_workload(model, ckpt, device, world, rank, lines_per_gpu, frames, batch_size):
  """The batch-synthesis configuration: world x lines_per_gpu synthetic lines, LPT-sharded over the ranks; returns
  (step function, frames of this rank per step, frames of the job per step, description)."""
  from tacotron_amd import scenarios, sharding
  from tacotron_amd.masks import prenet_keep_masks
  n_lines = world * lines_per_gpu
  all_lens = np.random.RandomState(2024).randint(20, 201, size=n_lines).tolist()
  shards = sharding.lpt_partition([sharding.predicted_cost(n) for n in all_lens], world)
  mine = sorted(shards[rank], key=lambda i: -all_lens[i])            # longest first: lines of a batch are alike
  masks = prenet_keep_masks(1111, frames, model.dims.P).to(device)
  batches = []
  for group in sharding.batches(mine, batch_size):
    lens = [all_lens[i] for i in group]
    ids = scenarios.random_ids(ckpt, lens, seed=2024 + group[0])
    d = lambda a: None if a is None else torch.from_numpy(a).to(device)   # noqa: E731
    batches.append(([d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None],
                    d(ids["speakers"]), d(ids["lens"])))

  def step():
    out = None
    for chans, spk, ln in batches:
      mem, pm = model.encode(chans, spk, ln)
      mel, gate, align, nf, reached = model.decode(mem, pm, ln, masks, frames, gate_threshold=2.0, want_align=False)
      out = model.postnet(mel, nf)
    return out

  desc = (f"{n_lines} synthetic lines (T_in ~ U{{20..200}}, seed 2024), {frames} decoder frames each (stop disabled), "
          f"partitioned longest-processing-time-first over {world} rank(s), padded batches of {batch_size} "
          "(lockstep decoder, one persistent launch per batch, 16 utterances in flight), encoder + decoder + postnet")
  return step, len(mine) * frames, n_lines * frames, desc


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=100)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--frames", type=int, default=1000, help="decoder frames per utterance (max_decoder_steps)")
  ap.add_argument("--workload", choices=("auto", "b1", "lines"), default="auto",
                  help="b1: one North-Wind utterance per step at batch 1 (the headline configuration); lines: the sharded batch-synthesis "
                       "line set; auto = b1 on one GPU, lines on several")
  ap.add_argument("--lines-per-gpu", type=int, default=256, help="lines workload: lines per rank and step")
  ap.add_argument("--batch-size", type=int, default=64, help="lines workload: utterances per padded batch")
  ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample size; 0 disables it")
  ap.add_argument("--rehearse-ranks", action="store_true",
                  help="no GPU work: every rank sleeps instead of synthesizing and the ranks meet over gloo; "
                       "exercises the multi-process launch, barrier and max-over-ranks plumbing on a CPU-only machine "
                       "(the printed line is marked rehearsal and is not a measurement)")
  args = ap.parse_args()

  if "WORLD_SIZE" not in os.environ and args.gpus > 1:
    # started plainly: this process has made no GPU call yet and makes none - it starts the N ranks as children
    # (torch.distributed.run, one process per GPU), passes their output through and ends with their exit status
    return spawn_ranks(args.gpus)
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
  if args.rehearse_ranks:
    return rehearse(args, world, rank)
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X; there is no CPU path (the cpu_baseline leg is only a side measurement)")
  torch.cuda.set_device(local_rank)
  device = torch.device("cuda", local_rank)
  dist = None
  barrier_backend = None
  if world > 1:
    import torch.distributed as dist
    # The ranks meet only at the timing barriers (no collective on the data path): RCCL, bound to the rank's GPU; if
    # its set-up fails on this node the same two calls go over gloo on the host instead of losing the measurement
    # (every rank takes the same branch: the failure of a collective set-up is collective)
    try:
      dist.init_process_group("nccl", device_id=device)
      probe = torch.zeros(1, device=device)
      dist.all_reduce(probe)
      torch.cuda.synchronize(device)
      barrier_backend = "nccl"
    except Exception as ex:  # pylint: disable=broad-except
      print(f"bench.py: RCCL set-up failed on rank {rank} ({type(ex).__name__}: {ex}); the timing barriers use gloo", file=sys.stderr, flush=True)
      if dist.is_initialized():
        dist.destroy_process_group()
      dist.init_process_group("gloo")
      barrier_backend = "gloo"

  from tacotron_amd import abi, scenarios
  from tacotron_amd.checkpoint import get_hparams
  from tacotron_amd.masks import prenet_keep_masks
  from tacotron_amd.model import Tacotron2

  ckpt = scenarios.checkpoint("ljs")
  hp = get_hparams(ckpt)
  model = Tacotron2(hp, **abi.mapping_counts(ckpt, hp))
  model.load_state_dict(ckpt["model_state"])
  model.to(device)
  workload = args.workload if args.workload != "auto" else ("b1" if world == 1 else "lines")
  frames = args.frames
  never = 2.0                                    # sigmoid(gate) never exceeds it

  def barrier():
    torch.cuda.synchronize(device)
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize(device)

  def max_over_ranks(seconds):
    if dist is None:
      return seconds
    t = torch.tensor([seconds], device=device if barrier_backend == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

  hop = hp.hop_length
  if workload == "lines":
    step, my_frames, job_frames, desc = lines_workload(model, ckpt, device, world, rank, args.lines_per_gpu, frames,
                                                       args.batch_size)
    for _ in range(args.warmup):
      step()
    rank_s = []

    def timed_step():
      t0 = time.perf_counter()
      step()
      torch.cuda.synchronize(device)
      rank_s.append(time.perf_counter() - t0)

    elapsed = timed_region(args.steps, timed_step, barrier, max_over_ranks)
    value = job_frames * args.steps / elapsed
    audio_s = job_frames / frames * (frames - 1) * hop / 22050.0
    out = {
      "metric": "mel frames/sec", "value": round(value, 1), "unit": "mel frames/s", "n_gpus": world,
      "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "config": {"workload": desc, "frames_per_step": job_frames, "utterances_per_step": job_frames // frames,
                 "parallelism": f"{world} rank(s), one full model replica each, utterance-sharded, host-side gather only"},
      "real_time_factor": round(args.steps * audio_s / elapsed, 1),
      "rank0_frames_per_s": round(my_frames / float(np.mean(rank_s)), 1),
      "per_gpu_frames_per_s": round(value / world, 1),
    }
    if world > 1:
      # The same workload on ONE GPU, measured in this run: every rank decodes the one-rank line set (lines_per_gpu lines,
      # what `--gpus 1 --workload lines` runs) on its own GPU, same warm-up and step counts; the slowest rank's time counts.
      # efficiency = value / (n_gpus x that figure).  (The default one-GPU run, `--gpus 1`, is another workload - batch 1,
      # the headline configuration - and its value is NOT the denominator of this one.)
      step1, _, job1, desc1 = lines_workload(model, ckpt, device, 1, 0, args.lines_per_gpu, frames, args.batch_size)
      for _ in range(args.warmup):
        step1()
      elapsed1 = timed_region(args.steps, lambda: (step1(), torch.cuda.synchronize(device)), barrier, max_over_ranks)
      one_gpu = job1 * args.steps / elapsed1
      out["one_gpu_same_workload"] = {"workload": desc1, "value": round(one_gpu, 1), "unit": "mel frames/s",
                                      "ms_per_step": round(elapsed1 / args.steps * 1e3, 4), "steps": args.steps, "warmup": args.warmup,
                                      "note": "measured in this run on every rank's GPU at once (slowest rank), no sharding"}
      out["efficiency"] = round(value / (world * one_gpu), 4)
      out["barrier_backend"] = barrier_backend
      out["note"] = ("weak scaling of the sharded line set: efficiency = value / (n_gpus x one_gpu_same_workload.value), both measured in "
                     "this run.  `python bench.py --gpus 1` runs the batch-1 headline configuration, a different workload: its value "
                     "must not be used as this curve's one-GPU point (its line carries `lines_one_gpu` for that purpose)")
    # the dominant kernel's roofline: measured on this rank's GPU outside the timed region (every rank does the same
    # work, rank 0's figures are printed)
    out["batch64"] = batch64_record(model, ckpt, device)
    out["roofline"] = out["batch64"]["roofline"]
    out["cpu_baseline"] = None
    if rank == 0 and args.cpu_seconds > 0:
      # rank 0 only, while the other ranks wait at the closing barrier
      sym, st, spk = build_inputs(ckpt)
      out["cpu_baseline"] = cpu_baseline(ckpt, sym, st, spk, args.cpu_seconds)
      out["cpu_baseline"]["sample"] += "; per frame the port's cost does not depend on how lines are batched"
    barrier()
    if rank == 0:
      print(json.dumps(out), flush=True)
    if dist is not None:
      dist.destroy_process_group()
    return

  sym, st, spk = build_inputs(ckpt)
  sym_d, st_d, spk_d = (torch.from_numpy(a).to(device) for a in (sym, st, spk))
  masks = prenet_keep_masks(1111, frames, model.dims.P).to(device)

  def step():
    mem, pm = model.encode([sym_d, st_d], spk_d, None)
    mel, gate, align, n_frames, reached = model.decode(mem, pm, None, masks, frames, gate_threshold=never,
                                                       want_align=True)
    post = model.postnet(mel, n_frames)
    return post, n_frames

  for _ in range(args.warmup):
    post, n_frames = step()
  assert int(n_frames[0]) == frames
  dec_us = []

  def timed_step():
    step()
    dec_us.append(model.last_decode_timing()[0])   # decode has completed already: the call returns at once

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
  us_per_frame = dec_s / frames * 1e6
  streaming = bytes_per_frame * frames / dec_s / 1e9
  audio_s = (frames - 1) * hop / 22050.0
  # HBM bytes of the decoder kernel and its issue / LDS figures come from the committed counter profiles (rocprofv3
  # cannot run inside this process): labelled with their source, per frame, scaled to this launch's frame count
  traffic, traffic_source, counters = None, None, None
  try:
    src = latest_profile("r*_pmc_traffic.json")
    with open(src) as f:
      traffic = int(json.load(f)["hbm_bytes_per_frame"] * frames)
    traffic_source = os.path.relpath(src, ROOT)
  except (TypeError, OSError, KeyError, ValueError):
    traffic = None
  try:
    src = latest_profile("r*_pmc_sq.json")
    with open(src) as f:
      sq = json.load(f)
    dv = sq["derived"]
    # LDS traffic: wave-level LDS instructions x 64 lanes x 16 bytes at most (the frame loop's reads are ds_read_b128),
    # against the ~150 TB/s every CU reading LDS together reaches
    lds_gbs = dv["SQ_INSTS_LDS_per_frame"] * 64 * 16 / (us_per_frame * 1e-6) / 1e9
    counters = {"source": os.path.relpath(src, ROOT),
                "issue_fraction_of_wave_cycles": dv["SQ_ACTIVE_INST_ANY_per_WAVE_CYCLE"],
                "valu_issue_fraction": dv["SQ_ACTIVE_INST_VALU_per_WAVE_CYCLE"],
                "scalar_issue_fraction": dv["SQ_ACTIVE_INST_SCA_per_WAVE_CYCLE"],
                "lds_issue_fraction": dv["SQ_ACTIVE_INST_LDS_per_WAVE_CYCLE"],
                "lds_bank_conflict_fraction": dv["lds_bank_conflict_cycles_per_active_cycle"],
                "lds_gbs_upper_bound": round(lds_gbs, 1), "lds_peak_gbs": 150000.0,
                "lds_frac": round(lds_gbs / 150000.0, 4)}
  except (TypeError, OSError, KeyError, ValueError):
    counters = None
  hop_us, hop_source = hop_cost()
  floor_us = HOPS_PER_FRAME * hop_us if hop_us else None
  out = {
    "metric": "mel frames/sec", "value": round(value, 1), "unit": "mel frames/s", "n_gpus": world,
    "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
    "config": {"workload": f"LJS-sized synthetic checkpoint (E=517, M=534), North-Wind line T_in={T_in}, B=1, "
                           f"{frames} decoder frames per utterance (stop disabled), encoder + persistent decoder + postnet",
               "frames_per_step": frames, "utterances_per_step": 1, "parallelism": f"{world} independent replica(s)"},
    "real_time_factor": round(world * args.steps * audio_s / elapsed, 1),
    "decoder_us_per_frame": round(us_per_frame, 3),
    "decoder_step_latency_us": step_latency,
    "roofline": {"bound": "handoff-latency", "achieved": round(us_per_frame, 3), "peak": floor_us and round(floor_us, 3), "unit": "us/frame",
                 "frac": floor_us and round(floor_us / us_per_frame, 4), "traffic": traffic, "traffic_source": traffic_source,
                 "kernel": "taco_decoder_kernel", "kernel_ms_per_launch": round(dec_s * 1e3, 4),
                 "hops_per_frame": HOPS_PER_FRAME, "hop_us": hop_us, "hop_source": hop_source,
                 "residency_ratio": round(streaming / HBM_PEAK_GBS, 4),
                 "hbm_streaming": {"bytes_per_frame": int(bytes_per_frame), "achieved_gbs": round(streaming, 1),
                                   "peak_gbs": HBM_PEAK_GBS},
                 "counters": counters,
                 "note": "batch 1: the LSTM rows are register-resident (residency_ratio = algorithmic bytes per frame / time over "
                         "the HBM peak; above 1 = faster than streaming the weights from HBM could be), so the bound that applies "
                         "is the chain of dependent cross-CU hand-offs: frac = (hops x cost of one hop, as the stand-alone microbenchmark "
                         "behind hop_source measured it) / measured frame time"},
  }
  if rank == 0 and world == 1:
    out["batch64"] = batch64_record(model, ckpt, device)
    out["batch64_bf16"] = batch64_bf16_record(model, ckpt, device, out["batch64"])
    # the workload `--gpus N` runs for N > 1, on this one GPU: the figure a whole-node value is to be compared with
    # (N x this = ideal weak scaling; every rank of an N-GPU run decodes the same number of lines as this run does)
    lstep, _, ljob, ldesc = lines_workload(model, ckpt, device, 1, 0, args.lines_per_gpu, frames, args.batch_size)
    lstep()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(3):
      lstep()
    torch.cuda.synchronize(device)
    lsec = (time.perf_counter() - t0) / 3
    out["lines_one_gpu"] = {"workload": ldesc, "value": round(ljob / lsec, 1), "unit": "mel frames/s",
                            "ms_per_step": round(lsec * 1e3, 3), "steps": 3,
                            "note": "same as `python bench.py --workload lines`; runs with --gpus N > 1 measure this workload"}
    # ... and lines whose stop test decides (the calibrated stop token of the parity tests' checkpoint; symbols drawn as
    # scripts/bulk_once.py draws them), through the product's own batch path, Synthesizer.launch_batch / finish_batch:
    # lines end at different frames, slots refill, and every launch but the last leaves its running lines to the next
    try:
      from tacotron_amd.synthesizer import Synthesizer
      with open(os.path.join(ROOT, "tests", "golden", "gate_bias.json")) as f:
        ckpt_stop = scenarios.checkpoint("ljs", json.load(f)["ljs"])
      synth = Synthesizer(ckpt_stop, device=device)
      keys = sorted(k for k in ckpt_stop["symbol_mapping"].keys() if k and "|" not in k and k.strip() == k)
      r = np.random.RandomState(2024)
      text = [[keys[i] for i in r.randint(0, len(keys), size=int(t))] for t in r.randint(20, 201, size=4 * args.lines_per_gpu)]
      order = sorted(range(len(text)), key=lambda i: -len(text[i]))
      groups = [order[i:i + args.batch_size] for i in range(0, len(order), args.batch_size)]
      speaker = next(iter(ckpt_stop["speaker_mapping"]))
      preps = [synth.prepare_batch([text[i] for i in g], speaker) for g in groups]

      def sstep():
        produced, carry = 0, None
        for g, grp in enumerate(groups):
          pend = synth.launch_batch([text[i] for i in grp], speaker, frames, 1111, prepared=preps[g], tags=grp, carry=carry,
                                    leave_unfinished=g + 1 < len(groups))
          carry = pend.unfinished
          produced += sum(res.mel_outputs_postnet.shape[1] for res in synth.finish_batch(pend))
          pend.release()
        return produced

      sstep()
      torch.cuda.synchronize(device)
      t0 = time.perf_counter()
      produced = sum(sstep() for _ in range(2))
      torch.cuda.synchronize(device)
      ssec = time.perf_counter() - t0
      out["lines_one_gpu_live_stop"] = {
        "workload": f"{len(text)} synthetic lines (T_in ~ U{{20..200}}), stop test live (calibrated stop token), at most {frames} decoder "
                    f"frames each, padded batches of {args.batch_size} through Synthesizer.launch_batch / finish_batch (host logic and "
                    "copies to the host included, symbol mapping done beforehand, nothing written to disk); every launch but the last "
                    "leaves the lines still running to the next one (taco_decode_resumable)",
        "value": round(produced / ssec, 1), "unit": "mel frames/s", "frames_per_step": produced // 2,
        "ms_per_step": round(ssec / 2 * 1e3, 3), "steps": 2}
      synth.model.close()
    except (OSError, KeyError, ValueError) as ex:
      out["lines_one_gpu_live_stop"] = {"skipped": str(ex)}
  if rank == 0 and world == 1 and args.cpu_seconds > 0:
    out["cpu_baseline"] = cpu_baseline(ckpt, sym, st, spk, args.cpu_seconds)
    try:
      z = np.load(os.path.join(ROOT, "tests", "golden", "ljs_1000.npz"))
      meta = json.loads(bytes(z["meta_json"]).decode("utf-8"))
      out["cpu_baseline"]["reference_container_ms_per_frame"] = round(meta["ref_seconds"] / meta["n_frames"] * 1e3, 4)
      out["cpu_baseline"]["reference_container_note"] = (
        f"the reference itself (PyTorch {meta['torch']}, {meta.get('ref_threads', 8)} threads) on this workload's 1000 frames "
        "in the build container, recorded with the golden tests/golden/ljs_1000.npz; it cannot run on the GPU box")
    except (OSError, KeyError, ValueError):
      pass
  elif rank == 0:
    out["cpu_baseline"] = None
  if rank == 0:
    print(json.dumps(out), flush=True)
  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
