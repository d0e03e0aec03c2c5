#!/usr/bin/env python3
"""Times the postnet stage on padded batches with ragged frame counts (what a bulk launch hands it): achieved
TFLOP/s over the frames that exist, against the same number of columns as one long utterance."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.model import Tacotron2

ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
r = np.random.RandomState(3)


def run(name, B, T_cap, nf):
  mel = torch.randn(B, 80, T_cap, device="cuda:0")
  nft = torch.tensor(nf, dtype=torch.int32, device="cuda:0")
  for _ in range(3):
    m.postnet(mel, nft)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  reps = 10
  e0.record()
  for _ in range(reps):
    m.postnet(mel, nft)
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / reps * 1000
  cols = int(np.sum(nf))
  flops = 2 * 5 * cols * (80 * 512 * 2 + 3 * 512 * 512)
  print(f"{name}: B={B} T_cap={T_cap}, {cols} frames: {us:.1f} us per call, {flops / us / 1e6:.1f} TFLOP/s over the frames that exist", flush=True)


run("one long utterance", 1, 34000, [34000])
run("64 x 1000, all full", 64, 1000, [1000] * 64)
ragged = r.randint(7, 1001, size=64)
run("64 ragged U{7..1000}", 64, 1000, list(ragged))
run("64 ragged, descending", 64, 1000, sorted(ragged, reverse=True))
run("64 ragged, 15 of them 0", 64, 1000, [0 if i < 15 else int(v) for i, v in enumerate(ragged)])
run("64 x 520", 64, 1000, [520] * 64)
run("64 x 520, T_cap 520", 64, 520, [520] * 64)
run("64 x 512, T_cap 512", 64, 512, [512] * 64)
