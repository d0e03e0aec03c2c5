import json, os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tacotron_amd") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, torch
from tacotron_amd import scenarios
from tacotron_amd.synthesizer import Synthesizer
gate = json.load(open(os.path.join(ROOT, "tests", "golden", "gate_bias.json")))["ljs"]
ckpt = scenarios.checkpoint("ljs", gate)
keys = sorted(k for k in ckpt["symbol_mapping"].keys() if k and "|" not in k and k.strip() == k)
r = np.random.RandomState(2024)
n_lines = 1280
lens = r.randint(20, 201, size=n_lines)
lines = [[keys[i] for i in r.randint(0, len(keys), size=int(t))] for t in lens]
s = Synthesizer(ckpt, device="cuda:0")
spk = next(iter(ckpt["speaker_mapping"]))
order = sorted(range(n_lines), key=lambda i: -len(lines[i]))
groups = [order[i:i + 64] for i in range(0, n_lines, 64)]
m = s.model
for rep in range(2):
  T = dict(prep=0, launch=0, decode=0, finish=0, total=0); frames = 0; lock_frames = 0
  t00 = time.perf_counter()
  import tacotron_amd.model as M
  orig = m.decode
  def timed_decode(*a, **k):
    t0 = time.perf_counter(); out = orig(*a, **k); T["decode"] += time.perf_counter() - t0
    return out
  m.decode = timed_decode
  T["encode"] = 0.0; T["postnet"] = 0.0; T["inference_batch"] = 0.0; T["to_host"] = 0.0
  o_enc, o_post, o_inf, o_th = m.encode, m.postnet, m.inference_batch, s._to_host
  def t_enc(*a, **k):
    t0 = time.perf_counter(); out = o_enc(*a, **k); T["encode"] += time.perf_counter() - t0; return out
  def t_post(*a, **k):
    t0 = time.perf_counter(); out = o_post(*a, **k); T["postnet"] += time.perf_counter() - t0; return out
  def t_inf(*a, **k):
    t0 = time.perf_counter(); out = o_inf(*a, **k); T["inference_batch"] += time.perf_counter() - t0; return out
  def t_th(*a, **k):
    t0 = time.perf_counter(); out = o_th(*a, **k); T["to_host"] += time.perf_counter() - t0; return out
  m.encode, m.postnet, m.inference_batch, s._to_host = t_enc, t_post, t_inf, t_th
  for g in groups:
    t0 = time.perf_counter(); preps = s.prepare_batch([lines[i] for i in g], spk); T["prep"] += time.perf_counter() - t0
    t0 = time.perf_counter(); p = s.launch_batch([lines[i] for i in g], spk, 1000, 1111, prepared=preps); T["launch"] += time.perf_counter() - t0
    us, lf = m.last_decode_timing(); lock_frames += lf; T.setdefault("dec_us", 0); T["dec_us"] += us
    t0 = time.perf_counter(); res = s.finish_batch(p); p.release(); T["finish"] += time.perf_counter() - t0
    frames += sum(x.mel_outputs_postnet.shape[1] for x in res)
  m.decode = orig
  m.encode, m.postnet, m.inference_batch, s._to_host = o_enc, o_post, o_inf, o_th
  T["total"] = time.perf_counter() - t00
  print({k: round(v, 3) for k, v in T.items()}, "frames", frames, "lockstep frames", lock_frames, "occupancy", round(frames / 16 / lock_frames, 3),
        "us per lockstep frame", round(T["dec_us"] / lock_frames, 2), "frames/s", round(frames / T["total"]))
