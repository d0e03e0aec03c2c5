#!/usr/bin/env python3
"""Throughput of a padded batch (default: 64 synthetic lines of 40..160 symbols, generator seed 2024, sorted by
length, every utterance forced to run `frames` decoder steps) through encode + decode + postnet.
TACO_PRECISION=bf16 switches the lockstep decoder to its bf16 setting."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 600
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
if os.environ.get("TACO_PRECISION"):
  m.set_precision(os.environ["TACO_PRECISION"])
lens = sorted(np.random.RandomState(2024).randint(40, 161, size=B).tolist(), reverse=True)
ids = scenarios.random_ids(ckpt, lens, seed=2024)
d = lambda a: None if a is None else torch.from_numpy(a).to("cuda:0")
chans = [d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None]
spk, ln = d(ids["speakers"]), d(ids["lens"])
masks = prenet_keep_masks(1111, frames, m.dims.P).to("cuda:0")

def run():
  mem, pm = m.encode(chans, spk, ln)
  mel, gate, align, nf, reached = m.decode(mem, pm, ln, masks, frames, gate_threshold=2.0, want_align=False)
  return m.postnet(mel, nf), nf

run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
  post, nf = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
total = int(nf.sum())
us, _ = m.last_decode_timing()
print(f"[{getattr(m, 'precision', 'fp32')}] B={B}, T_in {min(lens)}..{max(lens)}, {frames} frames each: {dt * 1e3:.1f} ms per batch = {total / dt:.0f} frames/s "
      f"({dt / B * 1e3:.2f} ms per utterance; decoder launches {us / 1e3:.1f} ms = {us / total:.2f} us/frame)")
