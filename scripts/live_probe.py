#!/usr/bin/env python3
"""Where the time of the live-stop line workload goes (bench.py's lines_one_gpu_live_stop: bulk-style lines with the
calibrated stop token through Synthesizer.launch_batch / finish_batch, unfinished lines carried into the next launch):
per launch the wall time of launch_batch, the decoder launch's device time inside it, and finish_batch.
Usage: live_probe.py [lines=1024] [batch_size=64] [max_decoder_steps=1000]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tacotron_amd import scenarios
from tacotron_amd.synthesizer import Synthesizer

n_lines = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
gate = json.load(open(os.path.join(ROOT, "tests", "golden", "gate_bias.json")))["ljs"]
ckpt = scenarios.checkpoint("ljs", gate)
synth = Synthesizer(ckpt, device="cuda:0")
keys = sorted(k for k in ckpt["symbol_mapping"].keys() if k and "|" not in k and k.strip() == k)
r = np.random.RandomState(2024)
text = [[keys[i] for i in r.randint(0, len(keys), size=int(t))] for t in r.randint(20, 201, size=n_lines)]
order = sorted(range(len(text)), key=lambda i: -len(text[i]))
groups = [order[i:i + bs] for i in range(0, len(order), bs)]
speaker = next(iter(ckpt["speaker_mapping"]))
preps = [synth.prepare_batch([text[i] for i in g], speaker) for g in groups]
m = synth.model
stage = {}
for name in ("encode", "decode", "postnet"):
  def wrap(fn, name=name):
    def timed(*a, **k):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      out = fn(*a, **k)
      torch.cuda.synchronize(); stage[name] = stage.get(name, 0.0) + time.perf_counter() - t0
      return out
    return timed
  if "--stages" in sys.argv:
    setattr(m, name, wrap(getattr(m, name)))

for rep in range(3):
  stage.clear()
  t_launch = t_finish = dec_dev = 0.0
  produced, carry = 0, None
  torch.cuda.synchronize(); t00 = time.perf_counter()
  for g, grp in enumerate(groups):
    t0 = time.perf_counter()
    pend = synth.launch_batch([text[i] for i in grp], speaker, steps, 1111, prepared=preps[g], tags=grp, carry=carry,
                              leave_unfinished=g + 1 < len(groups))
    t1 = time.perf_counter()
    dec_dev += m.last_decode_timing()[0] * 1e-6
    carry = pend.unfinished
    produced += sum(res.mel_outputs_postnet.shape[1] for res in synth.finish_batch(pend))
    pend.release()
    t2 = time.perf_counter()
    t_launch += t1 - t0; t_finish += t2 - t1
  torch.cuda.synchronize(); wall = time.perf_counter() - t00
  print(f"rep {rep}: {produced} frames, {len(groups)} launches, wall {wall * 1e3:.1f} ms = {produced / wall:.0f} frames/s; launch_batch {t_launch * 1e3:.1f} ms "
        f"(decoder on the device {dec_dev * 1e3:.1f} ms = {produced / dec_dev:.0f} frames/s), finish_batch {t_finish * 1e3:.1f} ms; per launch: "
        f"{t_launch / len(groups) * 1e3:.2f} ms, decoder {dec_dev / len(groups) * 1e3:.2f}, finish {t_finish / len(groups) * 1e3:.2f}"
        + ("; stages (synchronised): " + ", ".join(f"{k} {v * 1e3:.1f} ms" for k, v in stage.items()) if stage else ""), flush=True)
