#!/usr/bin/env python3
"""The bulk workload end to end on one GPU: `lines` synthetic lines (T_in ~ U{20..200}, generator seed 2024, symbols drawn
from the checkpoint's inventory) through the synthesize command in padded batches, on the LJS-dimension synthetic
checkpoint with its calibrated stop token, up to `max_decoder_steps` frames per line.  Reports wall-clock frames/s
(everything included: front-end, planning, encoder, decoder, postnet, the .npy files) and how the lines ended.
Usage: bulk_once.py [lines=10000] [max_decoder_steps=1000] [batch_size=64] [precision=fp32]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tacotron_amd import cli, scenarios

n_lines = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
precision = sys.argv[4] if len(sys.argv) > 4 else "fp32"
gate = json.load(open(os.path.join(ROOT, "tests", "golden", "gate_bias.json")))["ljs"]
ckpt = scenarios.checkpoint("ljs", gate)
keys = sorted(k for k in ckpt["symbol_mapping"].keys() if k and "|" not in k and k.strip() == k)
r = np.random.RandomState(2024)
lens = r.randint(20, 201, size=n_lines)
with tempfile.TemporaryDirectory() as tmp:
  ck_path = os.path.join(tmp, "ljs.pt")
  torch.save(ckpt, ck_path)
  text = os.path.join(tmp, "in.txt")
  with open(text, "w", encoding="UTF-8") as f:
    f.write("\n".join("|".join(keys[i] for i in r.randint(0, len(keys), size=int(t))) for t in lens))
  argv = ["synthesize", ck_path, text, "--sep", "|", "--custom-seed", "1111", "--max-decoder-steps", str(steps), "--batch-size", str(batch), "--precision", precision]
  cli.main(argv + ["--custom-lines", "1", "2", "3", "-out", os.path.join(tmp, "warm")])     # warm-up: library handle, masks
  t0 = time.perf_counter()
  rc = cli.main(argv + ["-out", os.path.join(tmp, "out")])
  wall = time.perf_counter() - t0
  files = [f for f in os.listdir(os.path.join(tmp, "out")) if f.endswith(".npy")]
  counts = np.array([np.load(os.path.join(tmp, "out", f), mmap_mode="r").shape[1] for f in files])
frames = int(counts.sum())
print(f"synthesize --batch-size {batch} --max-decoder-steps {steps} --precision {precision}: rc {rc}, {len(files)} of {n_lines} files, {frames} frames in {wall:.2f} s "
      f"= {frames / wall:.0f} frames/s wall ({wall / max(1, len(files)) * 1e3:.2f} ms per line); "
      f"{int((counts == steps).sum())} lines ran into the step limit, the others stopped after {int(counts[counts < steps].min()) if (counts < steps).any() else 0}.."
      f"{int(counts[counts < steps].max()) if (counts < steps).any() else 0} frames (median {int(np.median(counts))})")
