"""Host-side logic that needs no GPU: hparams handling, checkpoint format, symbol -> id mapping with
unmappable filtering, the C-ABI library's exported symbols."""
import ctypes
import dataclasses
import os
import re

import numpy as np
import pytest
import torch

from tacotron_amd import _lib, abi, checkpoint as ckpt_mod, config, scenarios, synth
from tacotron_amd.synthesizer import Synthesizer, spectrogram_duration_s
from conftest import ROOT, load_golden, scenario_checkpoint


def test_custom_hparams_types_and_errors():
  hp = config.HParams()
  out = config.apply_custom_hparams(hp, {"gate_threshold": "0.25", "prenet_dim": "128", "speakers_embedding_dim": ""})
  assert out.gate_threshold == 0.25 and out.prenet_dim == 128 and out.speakers_embedding_dim is None
  assert config.apply_custom_hparams(hp, None) is hp
  with pytest.raises(Exception):
    config.apply_custom_hparams(hp, {"no_such_param": "1"})
  assert config.parse_hparams_string("a=1,b=x") == {"a": "1", "b": "x"}
  assert config.parse_hparams_string(None) is None


def test_unknown_checkpoint_hparams_are_dropped():
  d = dataclasses.asdict(config.HParams())
  d["from_the_future"] = 3
  hp, dropped = config.hparams_from_dict(d)
  assert dropped == {"from_the_future"} and hp == config.HParams()


def test_model_dims_default_and_thchs():
  d = config.model_dims(config.HParams(), 80, 5, 2, None, None)
  assert (d.E, d.H_e, d.S, d.M) == (517, 259, 16, 534)
  assert d.step_params() == 18371975          # weights touched per decoder frame at default extents
  t = config.model_dims(synth.thchs_hparams(), 41, None, 61, 7, 5)
  assert (t.E, t.H_e, t.M) == (524, 262, 540)


def test_checkpoint_roundtrip(tmp_path, gate_table):
  ck = scenario_checkpoint("tiny", gate_table["tiny"])
  path = tmp_path / "1.pt"
  ckpt_mod.save_checkpoint(ck, path)
  back = ckpt_mod.load_checkpoint(path)
  assert set(back.keys()) == {"hparams", "model_state", "iteration", "learning_rate", "symbol_mapping",
                              "speaker_mapping", "stress_mapping"}
  assert ckpt_mod.get_hparams(back) == ckpt_mod.get_hparams(ck)
  for k, v in ck["model_state"].items():
    assert torch.equal(v, back["model_state"][k])
  with pytest.raises(FileNotFoundError):
    ckpt_mod.load_checkpoint(tmp_path / "missing.pt")
  with pytest.raises(KeyError):
    ckpt_mod.validate({"hparams": {}})


def _synth_without_gpu(ck):
  """A Synthesizer with its mappings but no device model (the mapping logic needs none)."""
  s = object.__new__(Synthesizer)
  s.hparams = ckpt_mod.get_hparams(ck)
  s.symbol_mapping = ck["symbol_mapping"]
  s.stress_mapping = ck.get("stress_mapping") if s.hparams.use_stress_embedding else None
  s.tone_mapping = ck.get("tone_mapping") if s.hparams.use_tone_embedding else None
  s.duration_mapping = ck.get("duration_mapping") if s.hparams.use_duration_embedding else None
  s.speaker_mapping = ck.get("speaker_mapping") if s.hparams.use_speaker_embedding else None
  return s


def test_symbols_to_ids_match_reference_inputs(gate_table):
  for name, symbols, speaker in (("ljs", scenarios.NORTH_WIND_SYMBOLS, "Linda Johnson"),
                                 ("thchs", scenarios.thchs_line(), "A38")):
    z, meta = load_golden(name)
    s = _synth_without_gpu(scenario_checkpoint(name, gate_table[name]))
    prep = s._prepare(symbols, speaker)
    for k in ("symbols", "stresses", "tones", "durations"):
      want = z["in_" + k][0].tolist() if ("in_" + k) in z else None
      assert prep.ids[k] == want, (name, k)
    assert prep.speaker_id == int(z["in_speakers"][0, 0])
    assert prep.unmappable_symbols is None and prep.unmappable_stresses is None


def test_unmappable_positions_are_dropped(gate_table):
  z, meta = load_golden("ljs_synth_unmappable")
  s = _synth_without_gpu(scenario_checkpoint("ljs", gate_table["ljs"]))
  prep = s._prepare(meta["symbols"], meta["speaker"])
  assert sorted(prep.unmappable_symbols) == meta["unmappable_symbols"]
  assert prep.unmappable_stresses is None and meta["unmappable_stresses"] is None
  n_bad = sum(1 for sym in meta["symbols"] if sym in ("XX", "q̃"))
  assert len(prep.ids["symbols"]) == len(meta["symbols"]) - n_bad == z["alignments"].shape[1]
  with pytest.raises(KeyError):
    s._prepare(meta["symbols"], "nobody")


def test_duration_formula_matches_reference_record():
  z, meta = load_golden("ljs_synth")
  n = z["mel_postnet"].shape[1]
  assert spectrogram_duration_s(n, 1024, 256) == meta["duration_s"]
  assert spectrogram_duration_s(214, 1024, 256) == 2.4729251700680273   # the reference's component test


def test_struct_layout_matches_header():
  """abi.TacoDims / TacoWeights must have the size the C compiler gives the header's structs."""
  import subprocess, tempfile
  src = '#include <stdio.h>\n#include "taco.h"\nint main(){printf("%zu %zu\\n", sizeof(taco_dims), sizeof(taco_weights));return 0;}\n'
  with tempfile.TemporaryDirectory() as td:
    c = os.path.join(td, "s.c")
    open(c, "w").write(src)
    exe = os.path.join(td, "s")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
    a, b = subprocess.check_output([exe]).split()
  assert int(a) == ctypes.sizeof(abi.TacoDims) and int(b) == ctypes.sizeof(abi.TacoWeights)


def test_library_exports_every_declared_symbol():
  """include/taco.h is the contract: the built library must export each function it declares
  (loading it needs no GPU; nothing is computed here)."""
  header = open(os.path.join(ROOT, "include", "taco.h")).read()
  declared = set(re.findall(r"\b(taco_[a-z_]+)\s*\(", header))
  declared -= {"taco_handle_s"}
  assert declared == set(_lib.EXPORTED_SYMBOLS)
  L = _lib.lib()
  for name in sorted(declared):
    assert hasattr(L, name), name
  assert L.taco_version().startswith(b"taco-hip")
  assert L.taco_last_error() == b""


def test_model_refuses_cpu_device(gate_table):
  from tacotron_amd.model import Tacotron2
  ck = scenario_checkpoint("tiny", gate_table["tiny"])
  hp = ckpt_mod.get_hparams(ck)
  m = Tacotron2(hp, **abi.mapping_counts(ck, hp))
  m.load_state_dict(ck["model_state"])
  with pytest.raises(RuntimeError):
    m.to("cpu")
  with pytest.raises(RuntimeError):
    m.step_weight_bytes


def test_bind_weights_refuses_tables_of_another_shape(gate_table):
  """The library copies ch_rows x ch_width floats from each table pointer: a checkpoint whose table is shorter
  or narrower than the mappings call for must be refused on the host, as load_state_dict would refuse it
  (reference: src/tacotron/training.py:762-768)."""
  ck = scenario_checkpoint("tiny", gate_table["tiny"])
  hp = ckpt_mod.get_hparams(ck)
  counts = abi.mapping_counts(ck, hp)
  cdims, md = abi.build_dims(hp, **counts)
  cpu = torch.device("cpu")
  abi.bind_weights(hp, md, ck["model_state"], cpu, cdims)          # the checkpoint as it is: accepted
  short = dict(ck["model_state"])
  short["symbol_embeddings.weight"] = short["symbol_embeddings.weight"][:-1]
  with pytest.raises(ValueError):
    abi.bind_weights(hp, md, short, cpu, cdims)
  if "speakers_embeddings.weight" in ck["model_state"]:
    narrow = dict(ck["model_state"])
    narrow["speakers_embeddings.weight"] = narrow["speakers_embeddings.weight"][:, :-1]
    with pytest.raises(ValueError):
      abi.bind_weights(hp, md, narrow, cpu, cdims)
