#!/usr/bin/env python3
"""Soak run of the decoder's hand-off protocol: many launches over several input lengths, every
launch compared bit for bit with the first one of its length (a lost or stale hand-off word would
change frames or end in the kernel's give-up path).  Usage: soak.py [launches per length] [frames]
TACO_PRECISION=bf16 runs the lockstep part in the bf16 setting."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.masks import prenet_keep_masks
from tacotron_amd.model import Tacotron2

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
d = lambda a: None if a is None else torch.from_numpy(a).to("cuda:0")
masks = prenet_keep_masks(1111, frames, m.dims.P).to("cuda:0")
bad = 0
for T_in in (1, 30, 102, 256, 300, 512, 700, 1024):
  ids = scenarios.random_ids(ckpt, [T_in], seed=T_in)
  chans = [d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None]
  mem, pm = m.encode(chans, d(ids["speakers"]), None)
  for i in range(n):                          # the encoder's recurrence uses the same kind of hand-off
    mem_i, pm_i = m.encode(chans, d(ids["speakers"]), None)
    if not (torch.equal(mem_i, mem) and torch.equal(pm_i, pm)):
      bad += 1
      print(f"T_in={T_in} encode {i}: output differs from the first call", flush=True)
  first = None
  post0 = None
  t0 = time.perf_counter()
  for i in range(n):
    mel, gate, align, nf, reached = m.decode(mem, pm, None, masks, frames, gate_threshold=2.0)
    # the convolutions whose tiles several workgroups share meet their partial tiles through a counter: same check
    pcols = min(250, frames)
    post = m.postnet(mel[:, :, :pcols].contiguous(), torch.full((1,), pcols, dtype=torch.int32, device="cuda:0"))
    if post0 is None:
      post0 = post.clone()
    elif not torch.equal(post, post0):
      bad += 1
      print(f"T_in={T_in} postnet {i}: output differs from the first call", flush=True)
    if first is None:
      first = (mel.clone(), gate.clone(), align.clone())
      assert int(nf[0]) == frames and bool(torch.isfinite(mel).all())
    elif not (torch.equal(mel, first[0]) and torch.equal(gate, first[1]) and torch.equal(align, first[2])):
      bad += 1
      print(f"T_in={T_in} launch {i}: output differs from the first launch", flush=True)
  torch.cuda.synchronize()
  print(f"T_in={T_in}: {n} launches x {frames} frames identical={bad == 0} ({time.perf_counter() - t0:.1f} s)", flush=True)
# the lockstep decoder: ragged batches with the stop test live (slots refill at different frames), every launch compared
# bit for bit with the first of its kind; and the same lines in reverse order must give the same frames
gate = __import__("json").load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gate_bias.json")))["ljs"]
ckpt = scenarios.checkpoint("ljs", gate)
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
if os.environ.get("TACO_PRECISION"):
  m.set_precision(os.environ["TACO_PRECISION"])
for B, steps in ((64, 400), (17, 150), (200, 250)):
  lens = np.random.RandomState(B).randint(3, 209, size=B).tolist()
  ids = scenarios.random_ids(ckpt, lens, seed=B)
  for b in range(B):
    ids["speakers"][b, :] = ids["speakers"][b, 0]
  chans = [d(ids[k]) for k in ("symbols", "stresses", "tones", "durations") if ids[k] is not None]
  ln = d(np.array(lens, np.int32))
  mem, pm = m.encode(chans, d(ids["speakers"]), ln)
  mk = prenet_keep_masks(7, steps, m.dims.P).to("cuda:0")
  flip = torch.arange(B - 1, -1, -1, device="cuda:0")
  mem_r, pm_r, ln_r = mem[flip].contiguous(), pm[flip].contiguous(), ln[flip].contiguous()
  first = None
  t0 = time.perf_counter()
  reps = max(4, n // 6)
  for i in range(reps):
    if i % 2 == 0:
      mel, gt, align, nf, reached = m.decode(mem, pm, ln, mk, steps)
    else:
      mel, gt, align, nf, reached = (x[flip] for x in m.decode(mem_r, pm_r, ln_r, mk, steps))
    if first is None:
      first = (mel.clone(), gt.clone(), align.clone(), nf.clone())
      assert bool(torch.isfinite(mel).all()) and len(set(nf.cpu().tolist())) > 3
    elif not (torch.equal(mel, first[0]) and torch.equal(gt, first[1]) and torch.equal(align, first[2]) and torch.equal(nf, first[3])):
      bad += 1
      print(f"lockstep B={B} launch {i}: output differs from the first launch", flush=True)
  torch.cuda.synchronize()
  print(f"lockstep B={B}: {reps} launches (forwards / backwards), {int(first[3].sum())} frames each, identical={bad == 0} "
        f"({time.perf_counter() - t0:.1f} s)", flush=True)
  # the same batch in three calls, each but the last leaving its running utterances to the next (taco_decode_resumable):
  # every utterance must come out bit for bit as in the one-call launches above
  t0 = time.perf_counter()
  thirds = [list(range(0, B // 3)), list(range(B // 3, 2 * B // 3)), list(range(2 * B // 3, B))]
  for i in range(max(2, reps // 4)):
    carried = []          # (utterance, slot, frames so far, mel row, gate row, align row)
    done = 0
    for c, new in enumerate(thirds):
      rows = [x[0] for x in carried] + new
      idx = torch.tensor(rows, device="cuda:0")
      resume = torch.tensor([x[1] for x in carried] + [-1] * len(new), dtype=torch.int32, device="cuda:0")
      mel, gt, align, nf, reached, slots = m.decode(mem[idx].contiguous(), pm[idx].contiguous(), ln[idx].contiguous(), mk, steps,
                                                    resume_slots=resume if carried else None, leave_unfinished=c + 1 < len(thirds))
      for b, x in enumerate(carried):
        mel[b, :, :x[2]] = x[3][:, :x[2]]; gt[b, :x[2]] = x[4][:x[2]]; align[b, :x[2]] = x[5][:x[2]]
      nf_h, re_h, sl_h = nf.cpu().tolist(), reached.cpu().tolist(), slots.cpu().tolist()
      carried = []
      for b, u in enumerate(rows):
        if re_h[b] == 2:
          carried.append((u, sl_h[b], nf_h[b], mel[b].clone(), gt[b].clone(), align[b].clone()))
          continue
        done += 1
        if not (nf_h[b] == int(first[3][u]) and torch.equal(mel[b], first[0][u]) and torch.equal(gt[b], first[1][u])
                and torch.equal(align[b], first[2][u])):
          bad += 1
          print(f"lockstep B={B} in three calls, pass {i}: utterance {u} differs from the one-call launch", flush=True)
    assert done == B and not carried
  torch.cuda.synchronize()
  print(f"lockstep B={B}: in three calls with utterances left and resumed, identical={bad == 0} ({time.perf_counter() - t0:.1f} s)", flush=True)
print("soak", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
