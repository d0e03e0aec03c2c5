"""Symbol front-end pins: the reference's own unit tests restated
(src/tacotron_tests/utils_py/test_cut_string.py, src/tacotron_tests/frontend/main_py/test_tone_detection.py)
and the reference's get_map_keys outputs recorded in tests/golden/frontend.json."""
import json
import os

import pytest

from tacotron_amd import frontend
from conftest import GOLDEN


@pytest.mark.parametrize("s,cut,rest,removed", [
  ("", set(), "", ""),
  ("", {"X"}, "", ""),
  ("X", {""}, "X", ""),
  ("X", {"X"}, "", "X"),
  ("yXzX", {"X"}, "yz", "XX"),
  ("yXX", {"X"}, "y", "XX"),
  ("yYzX", {"X", "Y"}, "yz", "YX"),
  ("iˑ", {"ˑ"}, "i", "ˑ"),
  ("a\u0306", {"\u0306"}, "a", "\u0306"),
  ("a˘", {"˘"}, "a", "˘"),
  ("\u0103", {"\u0306"}, "\u0103", ""),        # precomposed a-breve is one code point: unchanged
])
def test_cut_string(s, cut, rest, removed):
  assert frontend.cut_string(s, cut) == (rest, removed)


@pytest.mark.parametrize("symbol,core,tone", [
  ("i˧˩˧", "i", "˧˩˧"),
  ("ai˩˧", "ai", "˩˧"),
  ("b", "b", "-"),
  ("b˧", "b", "˧"),
])
def test_split_tone(symbol, core, tone):
  assert frontend.split_tone(symbol) == (core, tone)


def test_split_duration_and_stress_examples():
  assert frontend.split_duration("tː") == ("t", "ː")
  assert frontend.split_duration("t") == ("t", "-")
  assert frontend.split_stress("ˈɪ", True) == ("ɪ", "1")
  assert frontend.split_stress("ˌi", True) == ("i", "2")
  assert frontend.split_stress("ə", True) == ("ə", "0")
  assert frontend.split_stress("SIL0", True) == ("SIL0", "-")
  assert frontend.split_stress("ˈaɪ˘", True) == ("aɪ˘", "1")   # appendix ignored
  assert frontend.split_stress("AH1", False) == ("AH", "1")
  assert frontend.split_stress("AH", False) == ("AH", "0")
  assert frontend.split_stress("B", False) == ("B", "-")


def test_get_map_keys_matches_reference_records():
  with open(os.path.join(GOLDEN, "frontend.json"), encoding="utf-8") as f:
    data = json.load(f)
  symbols = tuple(data["symbols"])
  assert len(data["records"]) == 16

  class Flags:
    pass

  for rec in data["records"]:
    hp = Flags()
    for k in ("symbols_are_ipa", "use_stress_embedding", "use_tone_embedding", "use_duration_embedding"):
      setattr(hp, k, rec[k])
    cores, stresses, tones, durations = frontend.get_map_keys(symbols, hp)
    assert list(cores) == rec["cores"]
    for got, want in ((stresses, rec["stresses"]), (tones, rec["tones"]), (durations, rec["durations"])):
      assert (None if got is None else list(got)) == want


def test_mapping_helpers():
  m = frontend.build_mapping(["b", "a", "b"])
  assert m == {"a": 1, "b": 2}
  assert frontend.get_mappings_count(m) == 3
  assert frontend.get_mapped_indices(["a", "zz", "b"], m) == [1, None, 2]
