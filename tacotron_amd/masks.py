"""Keep-masks of the two always-on prenet dropouts, drawn from the reference's RNG stream.

The reference's prenet applies ``F.dropout(p=0.5, training=True)`` after each of its two layers
even in eval mode (reference: src/tacotron/model.py:136-141), and ``Synthesizer.infer`` seeds
torch's CPU generator right before inference (src/tacotron/synthesizer.py:176,
src/tacotron/utils.py:440-451).  On the CPU path those are the only random draws of an
inference call: per decoder step, ``empty(1, P).bernoulli_(0.5)`` for layer 1, then for layer 2;
the layer output is ``relu(x W^T) * keep * 2``.  The kernels take the keep bits as an input, so
the results stay bit-reproducible and comparable with the reference for the same seed.

The generator itself is not re-implemented: the masks come from ``torch`` on the host.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

_bulk_equals_stepwise: Dict[int, bool] = {}


def _stepwise(seed: int, n_steps: int, width: int) -> torch.Tensor:
  gen = torch.Generator(device="cpu")
  gen.manual_seed(seed)
  out = torch.empty(n_steps, 2, width, dtype=torch.uint8)
  row = torch.empty(1, width, dtype=torch.float32)
  for t in range(n_steps):
    for layer in range(2):
      row.bernoulli_(0.5, generator=gen)
      out[t, layer] = row[0].to(torch.uint8)
  return out


def _bulk(seed: int, n_steps: int, width: int) -> torch.Tensor:
  gen = torch.Generator(device="cpu")
  gen.manual_seed(seed)
  flat = torch.empty(n_steps * 2 * width, dtype=torch.float32)
  flat.bernoulli_(0.5, generator=gen)
  return flat.to(torch.uint8).view(n_steps, 2, width)


def prenet_keep_masks(seed: int, n_steps: int, width: int) -> torch.Tensor:
  """uint8 [n_steps, 2, width] of 0/1 on the CPU: mask[t, l] multiplies prenet layer ``l`` at
  decoder step ``t``.  One draw of the whole block is used when this torch build produces the
  same bits that way as with one ``bernoulli_`` call per layer and step (checked once per width
  on a prefix), otherwise the calls are replayed one by one."""
  if width not in _bulk_equals_stepwise:
    probe = 6
    _bulk_equals_stepwise[width] = bool(
      torch.equal(_stepwise(12345, probe, width), _bulk(12345, probe, width)))
  if _bulk_equals_stepwise[width]:
    return _bulk(seed, n_steps, width)
  return _stepwise(seed, n_steps, width)
