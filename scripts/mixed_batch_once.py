#!/usr/bin/env python3
"""Decoder time of a padded batch in which one line is longer than 256 symbols and the others are short:
what the short lines pay for the batch's padded width.  Usage: mixed_batch_once.py [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
lens = [300, 60, 80, 100, 120, 140, 90, 110]
ids = scenarios.random_ids(ckpt, lens, seed=5)
d = lambda a: None if a is None else torch.from_numpy(a).to("cuda:0")
chans = [d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None]
lens_d = torch.tensor(lens, dtype=torch.int32, device="cuda:0")
masks = prenet_keep_masks(1111, frames, m.dims.P).to("cuda:0")
mem, pm = m.encode(chans, d(ids["speakers"]), lens_d)
best = None
for _ in range(4):
  m.decode(mem, pm, lens_d, masks, frames, gate_threshold=2.0)
  torch.cuda.synchronize()
  us, _ = m.last_decode_timing()
  best = us if best is None else min(best, us)
print(f"lens {lens}: decoder launches {best / 1e3:.2f} ms for {len(lens)} x {frames} frames = "
      f"{best / (len(lens) * frames):.2f} us/frame [{os.environ.get('TACO_HIP_LIB', 'product')}]")
