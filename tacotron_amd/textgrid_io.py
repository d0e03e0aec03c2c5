"""Minimal reader for Praat TextGrid files: the interval marks of one named tier.

The reference reads grids with the third-party ``textgrid`` package
(src/tacotron_cli/textgrid_inference.py:116-127: ``TextGrid().read(path, 16, encoding)``,
``getFirst(tier)``, ``interval.mark`` of every interval); that package is not part of this
environment, so this module restates what those three calls deliver, for both of Praat's text
formats (the long one with ``name = value`` lines and the short one with bare values).  Both
formats hold the same sequence of values - strings in double quotes (a quote inside a string is
doubled), numbers, and the ``<exists>`` flag - and differ only in the labels around them, so the
reader scans for values and ignores everything else.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Tuple, Union

Value = Union[str, float, bool]


class TextGridError(ValueError):
  pass


@dataclass
class Tier:
  tier_class: str                 # "IntervalTier" or "TextTier"
  name: str
  xmin: float
  xmax: float
  entries: List[Tuple[float, float, str]]   # (xmin, xmax, mark); point tiers: (time, time, mark)

  @property
  def marks(self) -> Tuple[str, ...]:
    return tuple(mark for _, _, mark in self.entries)


def _scan(text: str) -> List[Tuple[str, Value]]:
  """All values of the file in order, as ("s", str) / ("n", float) / ("f", bool)."""
  out: List[Tuple[str, Value]] = []
  i, n = 0, len(text)
  while i < n:
    ch = text[i]
    if ch == '"':
      j = i + 1
      parts = []
      while True:
        k = text.find('"', j)
        if k < 0:
          raise TextGridError("unterminated string")
        parts.append(text[j:k])
        if k + 1 < n and text[k + 1] == '"':     # doubled quote = one literal quote
          parts.append('"')
          j = k + 2
          continue
        i = k + 1
        break
      out.append(("s", "".join(parts)))
    elif ch == "[":                              # "item [2]:" - an index, not a value
      k = text.find("]", i)
      i = n if k < 0 else k + 1
    elif ch == "<":
      k = text.find(">", i)
      if k < 0:
        raise TextGridError("unterminated flag")
      out.append(("f", text[i:k + 1] == "<exists>"))
      i = k + 1
    elif ch.isdigit() or (ch in "+-" and i + 1 < n and text[i + 1].isdigit()):
      prev = text[i - 1] if i > 0 else " "
      j = i + 1
      while j < n and (text[j].isdigit() or text[j] in ".eE" or (text[j] in "+-" and text[j - 1] in "eE")):
        j += 1
      if not (prev.isalnum() or prev == "_"):
        try:
          out.append(("n", float(text[i:j])))
        except ValueError as ex:
          raise TextGridError(f"bad number {text[i:j]!r}") from ex
      i = j
    elif ch == "!":                              # Praat comment to the end of the line
      k = text.find("\n", i)
      i = n if k < 0 else k + 1
    else:
      i += 1
  return out


def parse_textgrid(text: str) -> List[Tier]:
  vals = _scan(text)
  pos = 0

  def take(kind: str) -> Value:
    nonlocal pos
    if pos >= len(vals) or vals[pos][0] != kind:
      found = "end of file" if pos >= len(vals) else repr(vals[pos][1])
      raise TextGridError(f"malformed TextGrid: expected a {'string' if kind == 's' else 'number' if kind == 'n' else 'flag'}, found {found}")
    pos += 1
    return vals[pos - 1][1]

  if take("s") != "ooTextFile" or take("s") != "TextGrid":
    raise TextGridError("not a TextGrid text file")
  take("n")
  take("n")
  if not take("f"):
    return []
  n_tiers = int(take("n"))
  tiers: List[Tier] = []
  for _ in range(n_tiers):
    tier_class = str(take("s"))
    name = str(take("s"))
    xmin, xmax = float(take("n")), float(take("n"))
    count = int(take("n"))
    entries = []
    for _ in range(count):
      if tier_class == "IntervalTier":
        a, b = float(take("n")), float(take("n"))
      else:
        a = b = float(take("n"))
      entries.append((a, b, str(take("s"))))
    tiers.append(Tier(tier_class, name, xmin, xmax, entries))
  return tiers


def read_textgrid(path: Union[str, Path], encoding: str = "UTF-8") -> List[Tier]:
  data = Path(path).read_bytes()
  if data[:2] in (b"\xff\xfe", b"\xfe\xff"):     # Praat writes UTF-16 when a label needs it
    text = data.decode("utf-16")
  else:
    text = data.decode(encoding)
    if text.startswith("﻿"):
      text = text[1:]
  return parse_textgrid(text)


def first_tier(tiers: List[Tier], name: str) -> Optional[Tier]:
  """First tier called ``name`` (what the reference's ``grid.getFirst(name)`` returns)."""
  for tier in tiers:
    if tier.name == name:
      return tier
  return None
