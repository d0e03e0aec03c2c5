#!/usr/bin/env python3
"""Wall time of the synthesize command over many lines on one GPU: what everything around the kernels
(parsing, planning, front-end, np.save) costs per utterance.  Usage: cli_once.py [lines] [max_decoder_steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import cli, scenarios

n_lines = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
if "TACO_WRITE_BEHIND" in os.environ:            # 0 = wait for every line's files before the next line (for comparison)
  cli.WRITE_BEHIND = int(os.environ["TACO_WRITE_BEHIND"])
ckpt = scenarios.checkpoint("ljs")
r = np.random.RandomState(7)
with tempfile.TemporaryDirectory() as tmp:
  ck_path = os.path.join(tmp, "ljs.pt")
  torch.save(ckpt, ck_path)
  symbols = list(scenarios.NORTH_WIND_SYMBOLS)
  lines = []
  for _ in range(n_lines):
    n = int(r.randint(40, 161))
    start = int(r.randint(0, max(1, len(symbols) - 1)))
    lines.append("|".join((symbols * 3)[start:start + n]))
  text = os.path.join(tmp, "in.txt")
  with open(text, "w", encoding="UTF-8") as f:
    f.write("\n".join(lines))
  argv = ["synthesize", ck_path, text, "--sep", "|", "--custom-seed", "1111", "--max-decoder-steps", str(steps)] + (["--batch-size", sys.argv[3]] if len(sys.argv) > 3 else []) + ["-o"]
  cli.main(argv[:-1] + ["--custom-lines", "1", "2", "3", "-out", os.path.join(tmp, "warm")])     # warm-up: handle, masks
  t0 = time.perf_counter()
  rc = cli.main(argv[:-1] + ["-out", os.path.join(tmp, "out")])
  wall = time.perf_counter() - t0
  files = [f for f in os.listdir(os.path.join(tmp, "out")) if f.endswith(".npy")]
  frames = sum(np.load(os.path.join(tmp, "out", f)).shape[1] for f in files)
print(f"synthesize: rc {rc}, {len(files)} files, {frames} frames in {wall:.3f} s = {frames / wall:.0f} frames/s "
      f"({wall / max(1, len(files)) * 1e3:.2f} ms per line)")
