#!/usr/bin/env python3
"""Encoder and postnet outputs of a fixed set of inputs (ragged batch, single lines of several lengths) into an .npz,
for comparing two builds of the library bit for bit (TACO_HIP_LIB selects the build).  Usage: conv_bits.py out.npz"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.model import Tacotron2

ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
out = {}
r = np.random.RandomState(1)
lens = r.randint(3, 201, size=64).tolist()
ids = scenarios.random_ids(ckpt, lens, seed=2)
chans = [d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None]
mem, pm = m.encode(chans, d(ids["speakers"]), d(np.array(lens, np.int32)))
out["mem64"], out["pm64"] = mem.cpu().numpy(), pm.cpu().numpy()
for T in (1, 30, 102, 256):
  i1 = scenarios.random_ids(ckpt, [T], seed=T)
  c1 = [d(i1[k]) for k in ("symbols", "stresses", "tones", "durations") if i1[k] is not None]
  mem1, pm1 = m.encode(c1, d(i1["speakers"]), None)
  out[f"mem_{T}"], out[f"pm_{T}"] = mem1.cpu().numpy(), pm1.cpu().numpy()
g = torch.Generator().manual_seed(4)
nf = r.randint(1, 1001, size=48).tolist()
mel = torch.randn(48, m.dims.n_mel, 1000, generator=g)
for b, n in enumerate(nf):
  mel[b, :, n:] = 0.0
out["post48"] = m.postnet(mel.to("cuda:0"), d(np.array(nf, np.int32))).cpu().numpy()
for T in (64, 250, 500, 1000, 4000):
  x = torch.randn(1, m.dims.n_mel, T, generator=g).to("cuda:0")
  out[f"post_{T}"] = m.postnet(x, d(np.array([T], np.int32))).cpu().numpy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
