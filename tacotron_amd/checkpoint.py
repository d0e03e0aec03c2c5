"""Reading (and writing) checkpoints in the reference's on-disk format.

A checkpoint is a ``torch.save``d ordered dict with the keys ``hparams`` (``asdict(HParams)``),
``model_state``, ``optimizer_state``, optional ``scheduler_state``, ``iteration``,
``learning_rate``, ``symbol_mapping`` and the optional ``speaker_mapping`` / ``stress_mapping`` /
``tone_mapping`` / ``duration_mapping`` (reference: src/tacotron/checkpoint_handling.py:19-56).
Inference reads ``hparams``, ``model_state`` and the mappings only.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any, Dict, Optional, Union

import torch

from tacotron_amd.config import HParams, hparams_from_dict

KEY_HPARAMS = "hparams"
KEY_MODEL_STATE = "model_state"
KEY_ITERATION = "iteration"
KEY_LEARNING_RATE = "learning_rate"
KEY_SYMBOL_MAPPING = "symbol_mapping"
KEY_SPEAKER_MAPPING = "speaker_mapping"
KEY_STRESS_MAPPING = "stress_mapping"
KEY_TONE_MAPPING = "tone_mapping"
KEY_DURATION_MAPPING = "duration_mapping"

CheckpointDict = Dict[str, Any]


def load_checkpoint(path: Union[str, Path], device: Union[str, torch.device] = "cpu",
                    trusted: bool = False) -> CheckpointDict:
  """``torch.load(path, map_location=device)`` as the reference's CLI does
  (src/tacotron_cli/io.py:118-122).  The safe loader (``weights_only=True``) is tried first: a
  checkpoint holds only dicts, strings, numbers and tensors, which it accepts.  ``trusted=True``
  allows falling back to full unpickling for a local file the caller vouches for."""
  path = Path(path)
  if not path.is_file():
    raise FileNotFoundError(f"Checkpoint '{path}' not found")
  try:
    ckpt = torch.load(path, map_location=device, weights_only=True)
  except Exception as ex:  # pylint: disable=broad-except
    if not trusted:
      raise RuntimeError(
        f"'{path}' was refused by the safe loader ({ex}); pass trusted=True for a local file you trust") from ex
    logging.getLogger(__name__).warning("safe loader refused the checkpoint, unpickling it fully")
    ckpt = torch.load(path, map_location=device, weights_only=False)
  validate(ckpt)
  return ckpt


def save_checkpoint(ckpt: CheckpointDict, path: Union[str, Path]) -> None:
  torch.save(ckpt, str(path))


def validate(ckpt: CheckpointDict) -> None:
  for key in (KEY_HPARAMS, KEY_MODEL_STATE, KEY_SYMBOL_MAPPING):
    if key not in ckpt:
      raise KeyError(f"checkpoint lacks '{key}'")


def get_hparams(ckpt: CheckpointDict) -> HParams:
  hp, _ = hparams_from_dict(ckpt[KEY_HPARAMS])
  return hp


def get_mapping(ckpt: CheckpointDict, key: str) -> Optional[Dict[str, int]]:
  return ckpt.get(key)
