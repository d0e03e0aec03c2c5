#!/usr/bin/env python3
"""How much of a decoder launch is not frames: launch duration by HIP events minus the sum of the
per-frame intervals of the kernel's frame clock (= start-up until the first frame's stamp + tail).
Usage: startup_once.py [frames ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2
import bench

ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
sym, st, spk = bench.build_inputs(ckpt)
d = lambda a: torch.from_numpy(a).to("cuda:0")
mem, pm = m.encode([d(sym), d(st)], d(spk), None)
for frames in [int(a) for a in sys.argv[1:]] or [2, 10, 100, 1000]:
  masks = prenet_keep_masks(1111, frames, m.dims.P).to("cuda:0")
  rows = []
  for rep in range(4):
    m.record_frame_clock(frames)
    m.decode(mem, pm, None, masks, frames, gate_threshold=2.0)
    torch.cuda.synchronize()
    us, fr = m.last_decode_timing()
    lat = m.frame_latencies_us(frames).numpy()
    rows.append((us, float(lat.sum()), float(np.median(lat)) if len(lat) else 0.0))
  m.record_frame_clock(0)
  us, s, med = rows[-1]
  print(f"{frames} frames: launch {us:.1f} us, frame intervals sum {s:.1f} us (median {med:.2f}), rest {us - s:.1f} us; "
        f"all runs rest: {[round(r[0] - r[1], 1) for r in rows]}")
