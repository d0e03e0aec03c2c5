#!/usr/bin/env python3
"""Soak run of the decoder's hand-off protocol: many launches over several input lengths, every
launch compared bit for bit with the first one of its length (a lost or stale hand-off word would
change frames or end in the kernel's give-up path).  Usage: soak.py [launches per length] [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
d = lambda a: None if a is None else torch.from_numpy(a).to("cuda:0")
masks = prenet_keep_masks(1111, frames, m.dims.P).to("cuda:0")
bad = 0
for T_in in (1, 30, 102, 256, 300, 512):
  ids = scenarios.random_ids(ckpt, [T_in], seed=T_in)
  chans = [d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None]
  mem, pm = m.encode(chans, d(ids["speakers"]), None)
  for i in range(n):                          # the encoder's recurrence uses the same kind of hand-off
    mem_i, pm_i = m.encode(chans, d(ids["speakers"]), None)
    if not (torch.equal(mem_i, mem) and torch.equal(pm_i, pm)):
      bad += 1
      print(f"T_in={T_in} encode {i}: output differs from the first call", flush=True)
  first = None
  t0 = time.perf_counter()
  for i in range(n):
    mel, gate, align, nf, reached = m.decode(mem, pm, None, masks, frames, gate_threshold=2.0)
    if first is None:
      first = (mel.clone(), gate.clone(), align.clone())
      assert int(nf[0]) == frames and bool(torch.isfinite(mel).all())
    elif not (torch.equal(mel, first[0]) and torch.equal(gate, first[1]) and torch.equal(align, first[2])):
      bad += 1
      print(f"T_in={T_in} launch {i}: output differs from the first launch", flush=True)
  torch.cuda.synchronize()
  print(f"T_in={T_in}: {n} launches x {frames} frames identical={bad == 0} ({time.perf_counter() - t0:.1f} s)", flush=True)
print("soak", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
