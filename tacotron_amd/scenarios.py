"""Named synthetic workloads shared by the golden generator, the tests and the benchmark.

No pretrained checkpoint is reachable offline, so every workload is a synthetic checkpoint in
the reference's format (``tacotron_amd.synth``) plus a deterministic input.  The stop token
of a scenario (weight seed and bias) is calibrated once, against the reference, by
``tools/make_golden.py`` and recorded in tests/golden/gate_bias.json; ``checkpoint(name, gate)``
rebuilds the very same checkpoint anywhere.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from tacotron_amd import frontend
from tacotron_amd.config import HParams
from tacotron_amd.synth import make_checkpoint, thchs_hparams, tiny_hparams

# Phonetic transcription of "The North Wind and the Sun were disputing which was the stronger,
# when a traveler came along wrapped in a warm cloak." - the input line of the reference's
# README example (README.md:120), 102 symbols separated by "|".
NORTH_WIND_LINE = (
  "ð|ʌ|SIL0|n|ˈɔ|ɹ|θ|SIL0|w|ˈɪ|n|d|SIL0|ˈæ|n|d|SIL0|ð|ʌ|SIL0|s|ˈʌ|n|SIL0|w|ɝ|SIL0|d|ɪ|s|p|j|ˈu|t|ɪ|ŋ|"
  "SIL0|h|w|ˈɪ|t͡ʃ|SIL0|w|ˈɑ|z|SIL0|ð|ʌ|SIL0|s|t|ɹ|ˈɔ|ŋ|ɝ|,|SIL1|h|w|ˈɛ|n|SIL0|ʌ|SIL0|t|ɹ|ˈæ|v|ʌ|l|ɝ|"
  "SIL0|k|ˈeɪ|m|SIL0|ʌ|l|ˈɔ|ŋ|SIL0|ɹ|ˈæ|p|t|SIL0|ɪ|n|SIL0|ʌ|SIL0|w|ˈɔ|ɹ|m|SIL0|k|l|ˈoʊ|k|.|SIL2")
NORTH_WIND_SYMBOLS: Tuple[str, ...] = tuple(NORTH_WIND_LINE.split("|"))

# Input line of the reference's component test (src/tacotron_tests/synthesizer_py/test_Synthesizer.py:19).
COMPONENT_TEST_LINE = (
  "ð|ˈɪ|s|SIL0|ˈɪ|z|SIL0|ə|SIL0|tː|ˈɛ|s|t|SIL0|ˈæ|b|?|SIL2|ə|n˘|d|SIL0|ˈaɪ˘|m|SIL0|ð|ˈɛr˘|SIL0|θ|ˈʌr|d˘|"
  "ˌi|-|wː|ˈʌː|nː|.|SIL2")
COMPONENT_TEST_SYMBOLS: Tuple[str, ...] = tuple(COMPONENT_TEST_LINE.split("|"))


class _Flags:
  """Minimal stand-in for the hparams flags the symbol splitter reads."""

  def __init__(self, hp: HParams):
    self.use_stress_embedding = hp.use_stress_embedding
    self.use_tone_embedding = hp.use_tone_embedding
    self.use_duration_embedding = hp.use_duration_embedding
    self.symbols_are_ipa = hp.symbols_are_ipa


def _ljs_symbol_names() -> List[str]:
  cores, _, _, _ = frontend.get_map_keys(NORTH_WIND_SYMBOLS + COMPONENT_TEST_SYMBOLS, _Flags(HParams()))
  return sorted(set(cores))


THCHS_TONES = ["˥", "˧˥", "˨˩˦", "˥˩", "˧", "-"]             # mapping keys of the tone channel
THCHS_DURATIONS = ["ː", "ˑ", "˘", "-"]                      # mapping keys of the duration channel
THCHS_SPEAKERS = [f"A{i}" for i in range(2, 62)]            # 60 speakers -> 61 table rows


def checkpoint(name: str, gate: Optional[Sequence[float]] = None):
  """name: 'tiny' | 'ljs' | 'thchs'; gate = (gate_seed, gate_bias) as recorded in
  tests/golden/gate_bias.json, or None for the uncalibrated stop token."""
  gs = None if gate is None else int(gate[0])
  gb = None if gate is None else float(gate[1])
  if name == "tiny":
    return make_checkpoint(tiny_hparams(), n_symbols=30, seed=3, gate_bias=gb, gate_seed=gs)
  if name == "ljs":
    return make_checkpoint(HParams(), n_symbols=79, seed=7, gate_bias=gb, gate_seed=gs,
                           symbol_names=_ljs_symbol_names(), speaker_names=["Linda Johnson"])
  if name == "thchs":
    cores = [f"p{i}" for i in range(40)]
    return make_checkpoint(thchs_hparams(), n_symbols=40, n_tones=6, n_durations=4, n_speakers=60,
                           seed=11, gate_bias=gb, gate_seed=gs, symbol_names=cores, tone_names=THCHS_TONES,
                           duration_names=THCHS_DURATIONS, speaker_names=THCHS_SPEAKERS)
  raise KeyError(name)


def thchs_line(n: int = 57, seed: int = 21) -> Tuple[str, ...]:
  """Synthetic Chinese-style line: core symbol + optional tone marks + optional duration mark."""
  r = np.random.RandomState(seed)
  out = []
  for _ in range(n):
    core = f"p{r.randint(0, 40)}"
    tone = THCHS_TONES[r.randint(0, len(THCHS_TONES))]
    dur = THCHS_DURATIONS[r.randint(0, len(THCHS_DURATIONS))]
    out.append(core + ("" if tone == "-" else tone) + ("" if dur == "-" else dur))
  return tuple(out)


def random_ids(ckpt: Dict[str, object], lens: Sequence[int], seed: int) -> Dict[str, np.ndarray]:
  """Right-padded id matrices [B, max(lens)] for every channel the checkpoint models, ids drawn
  uniformly from 1..n-1 (0 is the padding id and fills the padding), one speaker id per row."""
  r = np.random.RandomState(seed)
  B, T = len(lens), int(max(lens))
  hp = ckpt["hparams"]
  out: Dict[str, np.ndarray] = {}

  def channel(key: str) -> np.ndarray:
    n = len(ckpt[key])
    a = np.zeros((B, T), dtype=np.int64)
    for b, ln in enumerate(lens):
      a[b, :ln] = r.randint(1, n + 1, size=ln)
    return a

  out["symbols"] = channel("symbol_mapping")
  out["stresses"] = channel("stress_mapping") if hp["use_stress_embedding"] else None
  out["tones"] = channel("tone_mapping") if hp["use_tone_embedding"] else None
  out["durations"] = channel("duration_mapping") if hp["use_duration_embedding"] else None
  if hp["use_speaker_embedding"]:
    n = len(ckpt["speaker_mapping"])
    spk = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
      spk[b, :] = r.randint(1, n + 1)
    out["speakers"] = spk
  else:
    out["speakers"] = None
  out["lens"] = np.asarray(lens, dtype=np.int32)
  return out
