#!/usr/bin/env python3
"""Development check of the lockstep decoder against the CPU checker (oracle): a ragged batch through
taco_decode (one lockstep launch for the lines of up to 224 symbols, sixteen in flight), each utterance compared
with the oracle's run of the same line alone.  Usage: lockstep_check.py [tiny|ljs|thchs] [B] [frames] [forced]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2
from oracle.oracle import OracleModel, build

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12
forced = len(sys.argv) > 4 and sys.argv[4] == "forced"
live_stop = len(sys.argv) > 4 and sys.argv[4] == "stop"
gate = json.load(open(os.path.join(ROOT, "tests", "golden", "gate_bias.json")))[name]
ck = scenarios.checkpoint(name, gate)
hp = get_hparams(ck)
build()
om = OracleModel(ck)
m = Tacotron2(hp, **abi.mapping_counts(ck, hp)); m.load_state_dict(ck["model_state"]); m.to("cuda:0")
r = np.random.RandomState(7)
hi = 200 if name != "tiny" else 256
lens = [int(v) for v in r.randint(3, hi + 1, size=B)]
ids = scenarios.random_ids(ck, lens, seed=77)
for b in range(B):
  ids["speakers"][b, :] = ids["speakers"][b, 0]
d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
chans = lambda x: [x[k] for k in ("symbols", "stresses", "tones", "durations") if x[k] is not None]
masks = prenet_keep_masks(5, frames, m.dims.P)
ln = d(np.array(lens, np.int32))
thr = None if live_stop else 2.0
mem, pm = m.encode([d(c) for c in chans(ids)], d(ids["speakers"]), ln)
ref = []
for b in range(B):
  one = {k: (None if v is None else v[b:b + 1, :lens[b]]) for k, v in ids.items() if k != "lens"}
  mem_o, pm_o = om.encode(chans(one), one["speakers"], None)
  ref.append(om.decode(mem_o, pm_o, None, masks.numpy(), frames, **({} if thr is None else {"gate_threshold": thr})))
fm = None
if forced:
  fm = torch.zeros(B, m.dims.n_mel, frames, device="cuda:0")
  for b in range(B):
    fm[b] = torch.from_numpy(ref[b][0][0])
t0 = time.perf_counter()
mel, gate_g, align, nf, reached = m.decode(mem, pm, ln, masks, frames, forced_mel=fm, **({} if thr is None else {"gate_threshold": thr}))
torch.cuda.synchronize()
print(f"decode returned in {time.perf_counter() - t0:.3f} s; frames {nf.cpu().tolist()}")
worst = 0.0
for b in range(B):
  mel_o, gate_o, align_o, nf_o, reached_o, _ = ref[b]
  n = int(nf_o[0])
  ok_n = int(nf[b]) == n and int(reached[b]) == int(reached_o[0])
  n = min(n, int(nf[b]))
  e_mel = float(np.abs(mel[b, :, :n].cpu().numpy() - mel_o[0, :, :n]).max())
  e_gate = float(np.abs(gate_g[b, :n].cpu().numpy() - gate_o[0, :n]).max())
  e_al = float(np.abs(align[b, :n, :lens[b]].cpu().numpy() - align_o[0, :n]).max())
  clean = not bool(mel[b, :, int(nf[b]):].any()) and not bool(align[b, :, lens[b]:].any())
  worst = max(worst, e_mel, e_gate, e_al)
  print(f"utt {b:2d} len {lens[b]:3d}: frames {int(nf[b])}/{int(nf_o[0])} {'ok' if ok_n else 'MISMATCH'}  mel {e_mel:.2e} gate {e_gate:.2e} align {e_al:.2e} padding {'clean' if clean else 'DIRTY'}")
  if e_mel > 1e-2:
    per = np.abs(mel[b, :, :n].cpu().numpy() - mel_o[0, :, :n]).max(axis=0)
    print("   per-frame mel error:", " ".join(f"{v:.1e}" for v in per[:16]))
    pa = np.abs(align[b, :n, :lens[b]].cpu().numpy() - align_o[0, :n]).max(axis=1)
    print("   per-frame align error:", " ".join(f"{v:.1e}" for v in pa[:16]))
print(f"worst error {worst:.3e}")
us, fr = m.last_decode_timing()
print(f"decoder: {us:.0f} us for {fr} lockstep frames = {us / max(fr, 1):.2f} us per lockstep frame")
