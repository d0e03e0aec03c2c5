#!/usr/bin/env python3
"""Wall time of Synthesizer.infer (host logic + encoder + decoder + postnet + copy back) per call, against
the device time of its decoder launch: what the Python boundary adds per utterance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import scenarios
from tacotron_amd.synthesizer import Synthesizer
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ckpt = scenarios.checkpoint("ljs")
s = Synthesizer(ckpt, device="cuda:0")
symbols = tuple(scenarios.NORTH_WIND_SYMBOLS)
speaker = next(iter(s.speaker_mapping))
for _ in range(3):
  r = s.infer(symbols, speaker, max_decoder_steps=steps, seed=1111)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
  r = s.infer(symbols, speaker, max_decoder_steps=steps, seed=1111)
wall = (time.perf_counter() - t0) / n
us, frames = s.model.last_decode_timing()
print(f"infer: {r.mel_outputs_postnet.shape[1]} frames, wall {wall * 1e3:.3f} ms per call, decoder launch {us / 1e3:.3f} ms "
      f"({frames} frames), reported inference_duration_s {r.inference_duration_s * 1e3:.3f} ms")
