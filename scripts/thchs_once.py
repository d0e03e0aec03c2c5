#!/usr/bin/env python3
"""Decoder time per frame of the THCHS-like configuration (tone and duration one-hot channels, 61 speakers with a
speaker embedding: E = 524, M = 540), 102 random symbols, 1000 frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2
ckpt = scenarios.checkpoint("thchs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
print("dims", m.dims.E, m.dims.M, m.dims.S)
ids = scenarios.random_ids(ckpt, [102], seed=5)
d = lambda a: torch.from_numpy(a).to("cuda:0")
ch = [d(ids[k]) for k in ("symbols","stresses","tones","durations") if ids[k] is not None]
frames = 1000
masks = prenet_keep_masks(1111, frames, m.dims.P).to("cuda:0")
ts = []
for _ in range(6):
  mem, pm = m.encode(ch, d(ids["speakers"]), None)
  out = m.decode(mem, pm, None, masks, frames, gate_threshold=2.0)
  post = m.postnet(out[0], out[3]); torch.cuda.synchronize()
  us, fr = m.last_decode_timing(); ts.append(us/fr)
print("thchs decode us/frame", sorted(ts[1:]))
