#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE (stefantaubert/tacotron, imported from
/root/reference with stand-ins for absent plotting/metric packages, see ref_import.py) on
synthetic checkpoints written by tacotron_amd.synth.

Runs only in the build container; the fixtures it writes are data (inputs + the reference's
outputs) and are what pins the CPU oracle and the HIP path to the reference.  Usage:

    python tools/make_golden.py [case ...]        # default: all cases
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.install()

import tacotron.model as ref_model  # noqa: E402
from tacotron.checkpoint_handling import get_hparams  # noqa: E402
from tacotron.frontend.main import get_map_keys as ref_get_map_keys  # noqa: E402
from tacotron.synthesizer import Synthesizer as RefSynthesizer  # noqa: E402
from tacotron.utils import init_global_seeds  # noqa: E402

from tacotron_amd import frontend, scenarios  # noqa: E402
from tacotron_amd.masks import prenet_keep_masks  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
N_TAPS = 3


def build_ref_model(ckpt):
  hp = get_hparams(ckpt)

  def count(key):
    return len(ckpt[key]) + 1 if key in ckpt else None

  m = ref_model.Tacotron2(hp, count("symbol_mapping"), count("stress_mapping"), count("speaker_mapping"),
                          count("tone_mapping"), count("duration_mapping"))
  m.load_state_dict(ckpt["model_state"])
  return m.eval()


def run_ref(model, ids, seed, max_steps, capture=True):
  """One B=1 reference inference; returns outputs + per-step taps + observed dropout pairs."""
  taps = {k: [] for k in ("prenet", "attention_hidden", "attention_cell", "context", "decoder_hidden",
                          "decoder_cell")}
  drops = []
  dec = model.decoder
  orig_decode = dec.decode
  orig_dropout = ref_model.F.dropout

  def decode(x):
    out = orig_decode(x)
    if capture and len(taps["prenet"]) < N_TAPS:
      taps["prenet"].append(x[0].clone())
      taps["attention_hidden"].append(dec.attention_hidden[0].clone())
      taps["attention_cell"].append(dec.attention_cell[0].clone())
      taps["context"].append(dec.attention_context[0].clone())
      taps["decoder_hidden"].append(dec.decoder_hidden[0].clone())
      taps["decoder_cell"].append(dec.decoder_cell[0].clone())
    return out

  def dropout(x, p=0.5, training=True, inplace=False):
    y = orig_dropout(x, p, training, inplace)
    if capture and training and p == 0.5 and len(drops) < 8:
      drops.append((x.clone(), y.clone()))
    return y

  dec.decode = decode
  ref_model.F.dropout = dropout
  try:
    def t(a):
      return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).long()
    init_global_seeds(seed)
    with torch.no_grad():
      mel, post, gate, align, reached = model.inference(
        t(ids["symbols"]), t(ids["stresses"]), t(ids["tones"]), t(ids["durations"]), t(ids["speakers"]),
        max_steps)
  finally:
    dec.decode = orig_decode
    ref_model.F.dropout = orig_dropout
  out = dict(mel=mel[0].numpy(), mel_postnet=post[0].numpy(), gate=gate[0, :, 0].numpy(),
             alignments=align[0].numpy(), reached=np.array(bool(reached)))
  if capture:
    for k, v in taps.items():
      out["tap_" + k] = torch.stack(v).numpy()
    # pin the RNG contract: dropout output == input * keep * 2 with keep from prenet_keep_masks
    masks = prenet_keep_masks(seed, 4, model.decoder.prenet.layers[1].linear_layer.weight.shape[0])
    for i, (x, y) in enumerate(drops):
      keep = masks[i // 2, i % 2].float()
      assert torch.equal(y[0], x[0] * keep * 2.0), "prenet mask stream does not match torch RNG contract"
    out["masks_head"] = masks.numpy()
  return out


def calibrate_gate(scen, model, ids, seed, lo, hi, horizon):
  """Runs the reference with the stop disabled, recording [decoder hidden (+) context] per frame;
  then searches stop-token weight seeds for the frame in [lo, hi] whose energy exceeds every
  earlier one by the widest margin, and puts the bias halfway.  The stop token feeds nothing
  back into the model, so one dry run serves every candidate."""
  from tacotron_amd.synth import _draw
  dec = model.decoder
  saved_threshold = dec.gate_threshold
  dec.gate_threshold = 2.0          # sigmoid never exceeds it
  hc = []
  orig_decode = dec.decode

  def decode(x):
    out = orig_decode(x)
    hc.append(torch.cat((dec.decoder_hidden[0], dec.attention_context[0])).double().numpy())
    return out

  dec.decode = decode
  try:
    run_ref(model, ids, seed, horizon, capture=False)
  finally:
    dec.decode = orig_decode
    dec.gate_threshold = saved_threshold
  hc = np.stack(hc)
  key = "decoder.gate_layer.linear_layer.weight"
  shape = tuple(dec.gate_layer.linear_layer.weight.shape)
  best = None
  for gate_seed in range(200):
    wv = _draw(gate_seed, key, shape, 1.0).double().numpy()[0]
    g = hc @ wv
    run_max = np.maximum.accumulate(g)
    for t in range(max(lo, 1), min(hi, len(g) - 1) + 1):
      margin = g[t] - run_max[t - 1]
      if best is None or margin > best[0]:
        best = (margin, gate_seed, t, -(g[t] + run_max[t - 1]) / 2.0)
  margin, gate_seed, t, bias = best
  assert margin > 0.02, margin
  return (gate_seed, float(np.float32(bias))), t + 1, float(margin)


def single_ids(ids, b):
  """Row b of a padded batch as an unpadded B=1 input."""
  ln = int(ids["lens"][b])
  return {k: (None if v is None else v[b:b + 1, :ln]) for k, v in ids.items() if k != "lens"}


def save(name, arrays, meta):
  os.makedirs(GOLDEN, exist_ok=True)
  arrays = {k: v for k, v in arrays.items() if v is not None}
  arrays["meta_json"] = np.frombuffer(json.dumps(meta, ensure_ascii=False).encode("utf-8"), dtype=np.uint8)
  path = os.path.join(GOLDEN, name + ".npz")
  np.savez_compressed(path, **arrays)
  print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)  {meta}")


def ids_from_symbols(ckpt, symbols, speaker):
  """Mapping ids for a symbol line using the REFERENCE's splitter (no unmappable symbols allowed)."""
  hp = get_hparams(ckpt)
  cores, stresses, tones, durations = ref_get_map_keys(tuple(symbols), hp)
  def ids(keys, mapkey):
    if keys is None:
      return None
    return np.array([[ckpt[mapkey][k] for k in keys]], dtype=np.int64)
  out = dict(symbols=ids(cores, "symbol_mapping"), stresses=ids(stresses, "stress_mapping"),
             tones=ids(tones, "tone_mapping"), durations=ids(durations, "duration_mapping"))
  T = out["symbols"].shape[1]
  out["speakers"] = np.full((1, T), ckpt["speaker_mapping"][speaker], dtype=np.int64) \
    if hp.use_speaker_embedding else None
  out["lens"] = np.array([T], dtype=np.int32)
  return out


def case_single(name, scen, ids, seed, lo, hi, horizon, max_steps, extra_meta=None):
  ckpt = scenarios.checkpoint(scen)
  model = build_ref_model(ckpt)
  bias, n_expected, margin = calibrate_gate(scen, model, single_ids(ids, 0), seed, lo, hi, horizon)
  ckpt = scenarios.checkpoint(scen, bias)
  model = build_ref_model(ckpt)
  t0 = time.perf_counter()
  out = run_ref(model, single_ids(ids, 0), seed, max_steps)
  dt = time.perf_counter() - t0
  n = out["mel"].shape[1]
  assert n == n_expected and not bool(out["reached"]), (n, n_expected)
  meta = dict(scenario=scen, gate=list(bias), seed=seed, max_steps=max_steps, n_frames=n, gate_margin=margin,
              ref_seconds=dt, torch=torch.__version__)
  meta.update(extra_meta or {})
  arrays = dict(out)
  for k, v in ids.items():
    if v is not None:
      arrays["in_" + k] = v
  save(name, arrays, meta)
  return ckpt, bias


def case_batch(name, scen, bias, lens, ids_seed, seed, max_steps):
  """N independent B=1 reference runs over the rows of one padded batch."""
  ckpt = scenarios.checkpoint(scen, bias)
  model = build_ref_model(ckpt)
  ids = scenarios.random_ids(ckpt, lens, ids_seed)
  arrays = {}
  n_frames, reached = [], []
  for b in range(len(lens)):
    out = run_ref(model, single_ids(ids, b), seed, max_steps, capture=False)
    for k in ("mel", "mel_postnet", "gate", "alignments"):
      arrays[f"{k}_{b}"] = out[k]
    n_frames.append(out["mel"].shape[1])
    reached.append(bool(out["reached"]))
  arrays["n_frames"] = np.array(n_frames, dtype=np.int32)
  arrays["reached"] = np.array(reached, dtype=np.uint8)
  for k, v in ids.items():
    if v is not None:
      arrays["in_" + k] = v
  save(name, arrays, dict(scenario=scen, gate=list(bias), seed=seed, max_steps=max_steps,
                          n_frames=n_frames, reached=reached, torch=torch.__version__))


def case_synthesizer(name, scen, bias, symbols, speaker, seed, max_steps):
  """Reference Synthesizer.infer on a symbol line (front-end + unmappable filtering + result record)."""
  ckpt = scenarios.checkpoint(scen, bias)
  synth = RefSynthesizer(ckpt, device=torch.device("cpu"))
  res = synth.infer(tuple(symbols), speaker, max_decoder_steps=max_steps, seed=seed, include_stats=True)
  def as_list(s):
    return None if s is None else sorted(s)
  meta = dict(scenario=scen, gate=list(bias), seed=seed, max_steps=max_steps, speaker=speaker,
              symbols=list(symbols), sampling_rate=res.sampling_rate,
              reached_max_decoder_steps=bool(res.reached_max_decoder_steps), duration_s=res.duration_s,
              unmappable_symbols=as_list(res.unmappable_symbols),
              unmappable_stresses=as_list(res.unmappable_stresses),
              unmappable_tones=as_list(res.unmappable_tones),
              unmappable_durations=as_list(res.unmappable_durations), torch=torch.__version__)
  save(name, dict(mel_postnet=res.mel_outputs_postnet, mel=res.mel_outputs, gate=res.gate_outputs,
                  alignments=res.alignments), meta)


def case_frontend(name):
  """Symbol-splitter pins: the reference's get_map_keys on a fixed symbol list for all 16 flag sets."""
  import itertools
  r = np.random.RandomState(99)
  alphabet = sorted(frontend.IPA_STRESSABLE) + list("ˈˌːˑ˘˥˦˧˨˩ʰ̃") + ["t", "d", "ɹ", "r", "SIL0", ".", "ŋ"]
  arpa = sorted(frontend.ARPA_VOWELS) + [v + s for v in sorted(frontend.ARPA_VOWELS) for s in "012"] + \
    ["B", "CH", "AA3", "X1", "1", ""]
  symbols = ["".join(alphabet[r.randint(len(alphabet))] for _ in range(r.randint(1, 5))) for _ in range(400)]
  symbols += [arpa[r.randint(len(arpa))] for _ in range(200)]
  symbols += list(scenarios.NORTH_WIND_SYMBOLS) + list(scenarios.COMPONENT_TEST_SYMBOLS)
  records = []
  for ipa, us, ut, ud in itertools.product((True, False), repeat=4):
    class HP:
      symbols_are_ipa, use_stress_embedding, use_tone_embedding, use_duration_embedding = ipa, us, ut, ud
    cores, st, to, du = ref_get_map_keys(tuple(symbols), HP)
    records.append(dict(symbols_are_ipa=ipa, use_stress_embedding=us, use_tone_embedding=ut,
                        use_duration_embedding=ud, cores=list(cores), stresses=None if st is None else list(st),
                        tones=None if to is None else list(to), durations=None if du is None else list(du)))
  path = os.path.join(GOLDEN, name + ".json")
  with open(path, "w", encoding="utf-8") as f:
    json.dump(dict(symbols=symbols, records=records), f, ensure_ascii=False)
  print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main(argv):
  want = set(argv) if argv else None

  def on(name):
    return want is None or name in want

  torch.set_num_threads(os.cpu_count())
  biases = {}
  bias_path = os.path.join(GOLDEN, "gate_bias.json")
  if os.path.exists(bias_path):
    biases = json.load(open(bias_path))

  if on("frontend"):
    case_frontend("frontend")

  if on("tiny"):
    ckpt = scenarios.checkpoint("tiny")
    ids = scenarios.random_ids(ckpt, [17], seed=101)
    _, biases["tiny"] = case_single("tiny", "tiny", ids, seed=5, lo=20, hi=40, horizon=64, max_steps=64)
    # the committed checkpoint file guards the generator against drift
    torch.save(scenarios.checkpoint("tiny", biases["tiny"]), os.path.join(GOLDEN, "tiny_checkpoint.pt"))
  if on("tiny_batch"):
    case_batch("tiny_batch", "tiny", biases["tiny"], lens=[23, 17, 12, 9, 5, 3, 2, 1], ids_seed=202, seed=5,
               max_steps=12)
  if on("ljs"):
    ckpt = scenarios.checkpoint("ljs")
    ids = ids_from_symbols(ckpt, scenarios.NORTH_WIND_SYMBOLS, "Linda Johnson")
    assert ids["symbols"].shape[1] == 102
    _, biases["ljs"] = case_single("ljs", "ljs", ids, seed=1111, lo=150, hi=220, horizon=240, max_steps=256,
                                   extra_meta=dict(speaker="Linda Johnson"))
  if on("ljs_synth"):
    case_synthesizer("ljs_synth", "ljs", biases["ljs"], scenarios.NORTH_WIND_SYMBOLS, "Linda Johnson", 1111, 256)
    odd = list(scenarios.COMPONENT_TEST_SYMBOLS[:20]) + ["XX", "ˈɪ", "q̃"] + list(scenarios.COMPONENT_TEST_SYMBOLS[20:])
    case_synthesizer("ljs_synth_unmappable", "ljs", biases["ljs"], odd, "Linda Johnson", 0, 40)
  if on("ljs_batch"):
    case_batch("ljs_batch", "ljs", biases["ljs"], lens=[61, 40, 33], ids_seed=303, seed=1111, max_steps=40)
  if on("ljs_1000"):
    # the benchmark's own configuration: the North-Wind line run to 1000 frames with the stop test switched off
    # (threshold above any sigmoid); mel, stop energies and every 50th alignment row are kept
    ckpt = scenarios.checkpoint("ljs", biases["ljs"])
    ids = ids_from_symbols(ckpt, scenarios.NORTH_WIND_SYMBOLS, "Linda Johnson")
    model = build_ref_model(ckpt)
    model.decoder.gate_threshold = 2.0
    t0 = time.perf_counter()
    out = run_ref(model, single_ids(ids, 0), 1111, 1000, capture=False)
    dt = time.perf_counter() - t0
    assert out["mel"].shape[1] == 1000 and bool(out["reached"])
    arrays = dict(mel=out["mel"], gate=out["gate"], alignments_every_50=out["alignments"][::50].copy(),
                  mel_postnet_every_50=out["mel_postnet"][:, ::50].copy())
    for k, v in ids.items():
      if v is not None:
        arrays["in_" + k] = v
    save("ljs_1000", arrays, dict(scenario="ljs", gate=list(biases["ljs"]), seed=1111, max_steps=1000, n_frames=1000,
                                  gate_threshold=2.0, align_stride=50, ref_seconds=dt, ref_threads=torch.get_num_threads(),
                                  speaker="Linda Johnson", torch=torch.__version__))
  if on("thchs"):
    ckpt = scenarios.checkpoint("thchs")
    ids = ids_from_symbols(ckpt, scenarios.thchs_line(), "A38")
    _, biases["thchs"] = case_single("thchs", "thchs", ids, seed=7, lo=60, hi=120, horizon=140, max_steps=160,
                                     extra_meta=dict(speaker="A38", symbols=list(scenarios.thchs_line())))
  if on("thchs_synth"):
    case_synthesizer("thchs_synth", "thchs", biases["thchs"], scenarios.thchs_line(), "A38", 7, 160)

  with open(bias_path, "w") as f:
    json.dump(biases, f, indent=1)
  print("gate biases:", biases)


if __name__ == "__main__":
  main(sys.argv[1:])
