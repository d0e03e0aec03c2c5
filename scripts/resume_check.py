#!/usr/bin/env python3
"""Development check of leaving / resuming: the same ragged lines decoded in launches of `bs` lines (each but the last
leaves its running lines to the next) and in one launch; reports every line whose frames differ, from which frame on,
and where it was left."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tacotron_amd import scenarios
from tacotron_amd.synthesizer import Synthesizer
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else 150
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
gate = json.load(open(os.path.join(ROOT, "tests", "golden", "gate_bias.json")))["ljs"]
ckpt = scenarios.checkpoint("ljs", gate)
keys = sorted(k for k in ckpt["symbol_mapping"].keys() if k and "|" not in k and k.strip() == k)
r = np.random.RandomState(99)
lens = r.randint(20, 201, size=n_lines)
lines = [[keys[i] for i in r.randint(0, len(keys), size=int(t))] for t in lens]
s = Synthesizer(ckpt, device="cuda:0")
spk = next(iter(ckpt["speaker_mapping"]))
order = sorted(range(n_lines), key=lambda i: -len(lines[i]))
one = s.infer_batch([lines[i] for i in order], spk, steps, 7, include_stats=True)
want = {i: res for i, res in zip(order, one)}
got, left_at = {}, {}
carry = None
groups = [order[i:i + bs] for i in range(0, n_lines, bs)]
for g, grp in enumerate(groups):
  p = s.launch_batch([lines[i] for i in grp], spk, steps, 7, include_stats=True, tags=grp, carry=carry, leave_unfinished=g + 1 < len(groups))
  carry = p.unfinished
  if carry:
    for tag, slot, nf in zip(carry.tags, carry.rows.slots, carry.rows.n_frames):
      left_at.setdefault(tag, []).append((g, slot, nf))
  for tag, res in zip(p.finished_tags, s.finish_batch(p)):
    res.mel_outputs_postnet = res.mel_outputs_postnet.copy(); res.mel_outputs = res.mel_outputs.copy(); res.alignments = res.alignments.copy()
    got[tag] = res
  p.release()
bad = 0
for i in range(n_lines):
  a, b = got[i], want[i]
  if a.mel_outputs.shape != b.mel_outputs.shape:
    print("line", i, "frames", a.mel_outputs.shape[1], "vs", b.mel_outputs.shape[1], "left", left_at.get(i)); bad += 1; continue
  d = np.abs(a.mel_outputs - b.mel_outputs).max(axis=0)
  da = np.abs(a.alignments - b.alignments).max(axis=1)
  if d.max() > 0 or da.max() > 0:
    bad += 1
    print("line", i, "len", len(lines[i]), "frames", a.mel_outputs.shape[1], "first mel diff at", int(np.argmax(d > 0)) if d.max() > 0 else None,
          "first align diff at", int(np.argmax(da > 0)) if da.max() > 0 else None, "max", float(d.max()), float(da.max()), "left", left_at.get(i))
print(f"{bad} of {n_lines} lines differ; {len(left_at)} lines were left at least once")
