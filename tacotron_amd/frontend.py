"""Symbol front-end: turns the symbols of one input line into per-channel mapping keys and ids.

A checkpoint trained with separate stress / tone / duration channels stores one id mapping
per channel; at synthesis time every input symbol is split into (core symbol, stress label,
tone label, duration label) in exactly that order, and a position whose key is missing from
any mapping is dropped from the utterance (reference: src/tacotron/frontend/main.py:16-132,
src/tacotron/frontend/stress_detection.py:44-85, src/tacotron/utils.py:62-70,
src/tacotron/synthesizer.py:98-148).  Id 0 is reserved for padding, so a mapping with n
entries gives an embedding table / one-hot block of n + 1 rows.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple

PADDING_SHIFT = 1
NA_LABEL = "-"
LABEL_UNSTRESSED, LABEL_PRIMARY, LABEL_SECONDARY = "0", "1", "2"

# --- IPA inventory (reference: src/tacotron/frontend/ipa_symbols.py) ---
_IPA_PRIMARY, _IPA_SECONDARY = "\u02c8", "\u02cc"
_IPA_VOWELS = set("iyɨᵻʉɯuɪʏʊeøɘɵɤoɛœɜɞʌɔæɐaɶɑɒ")
_IPA_SCHWAS = set("əɚɝ")
_IPA_DIPHTHONGS = set(
  "aʊ aɪ eɪ oʊ ɔɪ æʊ æɔ æɪ ʌʊ ʌɪ oɪ ʉu iʌ ɪər ɪə ɛər ɛɪ ɛə eə eər eʊ ʊər ʊə ʊu aɪə "
  "ər ɔr ɛr ɪr ʊr ʌr əɹ ɔɹ ɛɹ ɪɹ ʊɹ ʌɹ".split())
IPA_STRESSABLE: Set[str] = _IPA_VOWELS | _IPA_SCHWAS | _IPA_DIPHTHONGS

TONE_MARKERS: Set[str] = set("\u02e5\u02e6\u02e7\u02e8\u02e9")
DURATION_MARKERS: Set[str] = {"\u02d0", "\u02d1", "\u02d8"}   # long, half-long, extra-short
_IPA_TONES = set("\u00b9\u00b2\u00b3\u2074\u2075\u2076\u2077\u2078\u2079") | TONE_MARKERS \
  | {"\u030b", "\u0301", "\u0304", "\u0300", "\u030f"}
# Trailing marks ignored when deciding whether a symbol is a stressable vowel: long, half-long,
# extra-short (combining and spacing form), nasal, raised, syllabic, non-syllabic, glottalised,
# aspirated, breathy-voice aspirated, and every tone mark.
_IPA_APPENDIX = "".join(sorted(
  {"\u02d0", "\u02d1", "\u0306", "\u02d8", "\u0303", "\u031d", "\u0329", "\u032f", "\u02c0",
   "\u02b0", "\u02b1"} | _IPA_TONES))

# --- ARPAbet inventory (reference: src/tacotron/frontend/arpa_symbols.py) ---
ARPA_VOWELS: Set[str] = set("AA AE AH AO AW AX AXR AY EH ER EY IH IX IY OW OY UH UW UX".split())
_ARPA_DIGIT_LABEL = {"0": LABEL_UNSTRESSED, "1": LABEL_PRIMARY, "2": LABEL_SECONDARY}


def cut_string(s: str, cut: Set[str]) -> Tuple[str, str]:
  """(characters of ``s`` not in ``cut``, characters of ``s`` in ``cut``), both in order."""
  kept = "".join(ch for ch in s if ch not in cut)
  removed = "".join(ch for ch in s if ch in cut)
  return kept, removed


def split_stress_ipa(symbol: str) -> Tuple[str, str]:
  bare = symbol.rstrip(_IPA_APPENDIX)
  if bare in IPA_STRESSABLE:
    return symbol, LABEL_UNSTRESSED
  if len(bare) > 1 and bare[1:] in IPA_STRESSABLE:
    if bare[0] == _IPA_PRIMARY:
      return symbol[1:], LABEL_PRIMARY
    if bare[0] == _IPA_SECONDARY:
      return symbol[1:], LABEL_SECONDARY
  return symbol, NA_LABEL


def split_stress_arpa(symbol: str) -> Tuple[str, str]:
  if symbol in ARPA_VOWELS:
    return symbol, LABEL_UNSTRESSED
  if len(symbol) > 1 and symbol[-1] in _ARPA_DIGIT_LABEL and symbol[:-1] in ARPA_VOWELS:
    return symbol[:-1], _ARPA_DIGIT_LABEL[symbol[-1]]
  return symbol, NA_LABEL


def split_stress(symbol: str, is_ipa: bool) -> Tuple[str, str]:
  return split_stress_ipa(symbol) if is_ipa else split_stress_arpa(symbol)


def split_tone(symbol: str) -> Tuple[str, str]:
  core, tone = cut_string(symbol, TONE_MARKERS)
  return core, (tone if tone != "" else NA_LABEL)


def split_duration(symbol: str) -> Tuple[str, str]:
  core, duration = cut_string(symbol, DURATION_MARKERS)
  return core, (duration if duration != "" else NA_LABEL)


def get_map_keys(symbols: Sequence[str], hparams) -> Tuple[Tuple[str, ...], Optional[Tuple[str, ...]],
                                                          Optional[Tuple[str, ...]], Optional[Tuple[str, ...]]]:
  """Splits in the fixed order stress -> tone -> duration; a channel that the checkpoint does
  not model is ``None`` and its marks stay inside the core symbol."""
  cores = tuple(symbols)
  stresses = tones = durations = None
  if hparams.use_stress_embedding:
    pairs = [split_stress(s, hparams.symbols_are_ipa) for s in cores]
    cores, stresses = tuple(p[0] for p in pairs), tuple(p[1] for p in pairs)
  if hparams.use_tone_embedding:
    pairs = [split_tone(s) for s in cores]
    cores, tones = tuple(p[0] for p in pairs), tuple(p[1] for p in pairs)
  if hparams.use_duration_embedding:
    pairs = [split_duration(s) for s in cores]
    cores, durations = tuple(p[0] for p in pairs), tuple(p[1] for p in pairs)
  return cores, stresses, tones, durations


def get_mappings_count(mapping: Dict[str, int]) -> int:
  return len(mapping) + PADDING_SHIFT


def get_mapped_indices(keys: Iterable[str], mapping: Dict[str, int]) -> List[Optional[int]]:
  return [mapping.get(k, None) for k in keys]


def build_mapping(entries: Iterable[str]) -> Dict[str, int]:
  return {k: i for i, k in enumerate(sorted(set(entries)), start=PADDING_SHIFT)}
