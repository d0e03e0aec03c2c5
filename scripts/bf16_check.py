#!/usr/bin/env python3
"""How far the lockstep decoder's bf16 setting runs from its fp32 setting when nothing is forced: a batch of 64 ragged
lines on the LJS-sized synthetic checkpoint with its calibrated stop token; prints the largest frame difference up to
several frame indices and how many utterances stop at the same frame."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gate = json.load(open(os.path.join(root, "tests", "golden", "gate_bias.json")))["ljs"]
ckpt = scenarios.checkpoint("ljs", gate)
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
lens = np.random.RandomState(5).randint(20, 201, size=64).tolist()
ids = scenarios.random_ids(ckpt, lens, seed=5)
d = lambda a: None if a is None else torch.from_numpy(a).to("cuda:0")
chans = [d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None]
spk, ln = d(ids["speakers"]), d(ids["lens"])
frames = 1000
masks = prenet_keep_masks(7, frames, m.dims.P).to("cuda:0")
mem, pm = m.encode(chans, spk, ln)
out = {}
for prec in ("fp32", "bf16"):
  m.set_precision(prec)
  mel, gate_e, align, nf, reached = m.decode(mem, pm, ln, masks, frames)
  out[prec] = (mel.cpu().numpy(), nf.cpu().numpy())
a, na = out["fp32"]
b, nb = out["bf16"]
print(f"stop frames: {int((na == nb).sum())} of {len(na)} utterances identical; largest difference {int(np.abs(na - nb).max())} frames "
      f"(fp32 frames {int(na.min())}..{int(na.max())})")
for upto in (5, 10, 20, 50, 100, 200):
  worst = 0.0
  for u in range(len(na)):
    n = min(int(na[u]), int(nb[u]), upto)
    worst = max(worst, float(np.abs(a[u, :, :n] - b[u, :, :n]).max()))
  print(f"frames < {upto}: max |mel bf16 - mel fp32| = {worst:.2e}")
