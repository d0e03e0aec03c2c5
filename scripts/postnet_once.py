#!/usr/bin/env python3
"""Times the postnet stage (5 convolutions) of the LJS-sized synthetic checkpoint for several lengths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.model import Tacotron2

ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
for T in [int(a) for a in sys.argv[1:]] or [250, 1000, 4000]:
  mel = torch.randn(1, 80, T, device="cuda:0")
  nf = torch.tensor([T], dtype=torch.int32, device="cuda:0")
  for _ in range(3):
    m.postnet(mel, nf)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  reps = 20
  e0.record()
  for _ in range(reps):
    m.postnet(mel, nf)
  e1.record()
  torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / reps * 1000
  flops = 2 * 5 * T * (80 * 512 * 2 + 3 * 512 * 512)
  print(f"postnet T={T}: {us:.1f} us per call, {flops / us / 1e6:.1f} TFLOP/s [{os.environ.get('TACO_HIP_LIB', 'product')}]")
