#!/usr/bin/env python3
"""Runs the benchmark workload's decoder a few times (used with the stage-stamp diagnostic build and
under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2
import bench

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
sym, st, spk = bench.build_inputs(ckpt)
if len(sys.argv) > 3:                       # optional: synthetic input of that many symbols instead of the North-Wind line
  ids = scenarios.random_ids(ckpt, [int(sys.argv[3])], seed=77)
  sym, st, spk = ids["symbols"], ids["stresses"], ids["speakers"]
d = lambda a: torch.from_numpy(a).to("cuda:0")
masks = prenet_keep_masks(1111, frames, m.dims.P).to("cuda:0")
times = []
for _ in range(reps):
  mem, pm = m.encode([d(sym), d(st)], d(spk), None)
  out = m.decode(mem, pm, None, masks, frames, gate_threshold=2.0)
  post = m.postnet(out[0], out[3])
  torch.cuda.synchronize()
  us, fr = m.last_decode_timing()
  times.append(us / fr)
times = sorted(times[1:] or times)
print(f"T_in={sym.shape[1]} decode {fr} frames: min {times[0]:.2f} median {times[len(times) // 2]:.2f} max {times[-1]:.2f} us/frame "
      f"({len(times)} timed runs) [{os.environ.get('TACO_HIP_LIB', 'product')}]")
