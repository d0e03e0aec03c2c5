"""Deterministic synthetic checkpoints in the reference's on-disk format.

No pretrained checkpoint can be fetched offline, so tests, goldens and the benchmark use
checkpoints written by this generator: same dict keys, same ``model_state`` key names and
shapes as a checkpoint saved by the reference (reference: src/tacotron/checkpoint_handling.py:19-56,
src/tacotron/model.py:612-688).  Every tensor is drawn from ``numpy.random.RandomState`` seeded
by (base seed, crc32 of the key name): the legacy RandomState stream is frozen across numpy
versions, so the same call yields the same bytes in the build container and on the GPU box.
"""
from __future__ import annotations

import dataclasses
import zlib
from collections import OrderedDict
from math import sqrt
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from tacotron_amd.config import HParams, ModelDims, channel_widths, model_dims


def state_layout(hp: HParams, n_symbols: int, n_stresses: Optional[int], n_speakers: Optional[int],
                 n_tones: Optional[int], n_durations: Optional[int]) -> "OrderedDict[str, Tuple[int, ...]]":
  """``model_state`` key -> shape, in module registration order."""
  d = model_dims(hp, n_symbols, n_stresses, n_speakers, n_tones, n_durations)
  out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
  if hp.train_symbol_with_embedding:
    out["symbol_embeddings.weight"] = (n_symbols, hp.symbols_embedding_dim)
  if hp.use_speaker_embedding and hp.train_speaker_with_embedding:
    out["speakers_embeddings.weight"] = (n_speakers, hp.speakers_embedding_dim)
  if hp.use_stress_embedding and hp.train_stress_with_embedding:
    out["stress_embedding.weight"] = (n_stresses, hp.stress_embedding_dim)
  if hp.use_tone_embedding and hp.train_tone_with_embedding:
    out["tone_embedding.weight"] = (n_tones, hp.tone_embedding_dim)
  if hp.use_duration_embedding and hp.train_duration_with_embedding:
    out["duration_embedding.weight"] = (n_durations, hp.duration_embedding_dim)

  def conv_bn(prefix: str, c_in: int, c_out: int, k: int) -> None:
    out[f"{prefix}.0.conv.weight"] = (c_out, c_in, k)
    out[f"{prefix}.0.conv.bias"] = (c_out,)
    out[f"{prefix}.1.weight"] = (c_out,)
    out[f"{prefix}.1.bias"] = (c_out,)
    out[f"{prefix}.1.running_mean"] = (c_out,)
    out[f"{prefix}.1.running_var"] = (c_out,)
    out[f"{prefix}.1.num_batches_tracked"] = ()

  for i in range(d.n_enc):
    conv_bn(f"encoder.convolutions.{i}", d.E, d.E, d.k_enc)
  for suffix in ("", "_reverse"):
    out[f"encoder.lstm.weight_ih_l0{suffix}"] = (4 * d.H_e, d.E)
    out[f"encoder.lstm.weight_hh_l0{suffix}"] = (4 * d.H_e, d.H_e)
    out[f"encoder.lstm.bias_ih_l0{suffix}"] = (4 * d.H_e,)
    out[f"encoder.lstm.bias_hh_l0{suffix}"] = (4 * d.H_e,)

  out["decoder.prenet.layers.0.linear_layer.weight"] = (d.P, d.n_mel)
  out["decoder.prenet.layers.1.linear_layer.weight"] = (d.P, d.P)
  out["decoder.attention_rnn.weight_ih"] = (4 * d.R_a, d.P + d.M)
  out["decoder.attention_rnn.weight_hh"] = (4 * d.R_a, d.R_a)
  out["decoder.attention_rnn.bias_ih"] = (4 * d.R_a,)
  out["decoder.attention_rnn.bias_hh"] = (4 * d.R_a,)
  out["decoder.attention_layer.query_layer.linear_layer.weight"] = (d.A, d.R_a)
  out["decoder.attention_layer.memory_layer.linear_layer.weight"] = (d.A, d.M)
  out["decoder.attention_layer.v.linear_layer.weight"] = (1, d.A)
  out["decoder.attention_layer.location_layer.location_conv.conv.weight"] = (d.F, 2, d.K)
  out["decoder.attention_layer.location_layer.location_dense.linear_layer.weight"] = (d.A, d.F)
  out["decoder.decoder_rnn.weight_ih"] = (4 * d.R_d, d.R_a + d.M)
  out["decoder.decoder_rnn.weight_hh"] = (4 * d.R_d, d.R_d)
  out["decoder.decoder_rnn.bias_ih"] = (4 * d.R_d,)
  out["decoder.decoder_rnn.bias_hh"] = (4 * d.R_d,)
  out["decoder.linear_projection.linear_layer.weight"] = (d.n_mel, d.R_d + d.M)
  out["decoder.linear_projection.linear_layer.bias"] = (d.n_mel,)
  out["decoder.gate_layer.linear_layer.weight"] = (1, d.R_d + d.M)
  out["decoder.gate_layer.linear_layer.bias"] = (1,)

  for i in range(d.n_post):
    c_in = d.n_mel if i == 0 else d.C_post
    c_out = d.n_mel if i == d.n_post - 1 else d.C_post
    conv_bn(f"postnet.convolutions.{i}", c_in, c_out, d.k_post)
  return out


GATE_GAIN = 20.0


def _rng(seed: int, key: str) -> np.random.RandomState:
  return np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode("utf-8"))) % (2 ** 32))


def _draw(seed: int, key: str, shape: Tuple[int, ...], sharp: float) -> torch.Tensor:
  r = _rng(seed, key)
  if key.endswith("num_batches_tracked"):
    return torch.tensor(1000, dtype=torch.int64)
  if key.endswith(".1.weight"):                       # BatchNorm scale
    a = r.uniform(0.6, 1.4, size=shape)
  elif key.endswith(".1.running_var"):
    a = r.uniform(0.5, 1.5, size=shape)
  elif key.endswith(".1.bias") or key.endswith(".1.running_mean"):
    a = r.uniform(-0.2, 0.2, size=shape)
  elif len(shape) == 1:                               # conv / linear / LSTM bias
    a = r.uniform(-0.05, 0.05, size=shape)
  elif "lstm" in key or "_rnn." in key:               # LSTM matrices: +-1/sqrt(hidden)
    hidden = shape[0] // 4
    bound = 1.0 / sqrt(hidden)
    a = r.uniform(-bound, bound, size=shape)
  else:                                               # Glorot-uniform on (fan_out, fan_in[, k])
    receptive = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    bound = sqrt(6.0 / ((shape[0] + shape[1]) * receptive))
    if "attention_layer" in key:
      bound *= sharp                                  # peaked alignments instead of flat ones
    if "gate_layer" in key:
      bound *= GATE_GAIN                              # stop-token energies of order 0.1, not 0.003
    a = r.uniform(-bound, bound, size=shape)
  return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def make_mapping(prefix: str, n: int) -> "OrderedDict[str, int]":
  """``n`` entries with ids 1..n (id 0 is the padding id)."""
  return OrderedDict((f"{prefix}{i}", i) for i in range(1, n + 1))


def make_checkpoint(hp: Optional[HParams] = None, *, n_symbols: int = 79, n_stresses: int = 4,
                    n_tones: int = 6, n_durations: int = 4, n_speakers: int = 1, seed: int = 7,
                    sharp: float = 3.0, gate_bias: Optional[float] = None, gate_seed: Optional[int] = None,
                    symbol_names: Optional[Sequence[str]] = None,
                    tone_names: Optional[Sequence[str]] = None,
                    duration_names: Optional[Sequence[str]] = None,
                    speaker_names: Optional[Sequence[str]] = None) -> "OrderedDict[str, object]":
  """Checkpoint dict as the reference's ``checkpoint_handling.create`` lays it out, minus the
  optimizer state (an inference-only checkpoint, reference: checkpoint_handling.py:59-63, but
  keeping ``learning_rate`` because the CLI logs it).  ``n_*`` count mapping entries without
  the padding id; ``*_names`` give the entries explicit names (sorted, ids from 1, as the
  reference's ``build_mapping`` does: src/tacotron/frontend/main.py:194-199) and are topped up
  with filler names to ``n_*`` entries.  ``gate_bias`` / ``gate_seed`` override the stop-token
  bias and the seed its weight vector is drawn with: a random-weight model has no reason to stop
  at a useful frame, so goldens search a (seed, bias) pair that makes it stop where wanted; the
  stop token feeds nothing back, so this changes no other output."""
  hp = hp or HParams()

  def named(names, prefix, n):
    if names is None:
      return make_mapping(prefix, n)
    uniq = sorted(set(names))
    filler = [f"<{prefix}{i}>" for i in range(max(0, n - len(uniq)))]
    return OrderedDict((k, i) for i, k in enumerate(sorted(uniq + filler), start=1))

  symbol_mapping = named(symbol_names, "s", n_symbols)
  stress_mapping = OrderedDict((k, i) for i, k in enumerate(["-", "0", "1", "2"][:n_stresses], start=1)) \
    if hp.use_stress_embedding else None
  tone_mapping = named(tone_names, "t", n_tones) if hp.use_tone_embedding else None
  duration_mapping = named(duration_names, "d", n_durations) if hp.use_duration_embedding else None
  speaker_mapping = named(speaker_names, "speaker", n_speakers) if hp.use_speaker_embedding else None

  def count(m):
    return None if m is None else len(m) + 1

  layout = state_layout(hp, count(symbol_mapping), count(stress_mapping), count(speaker_mapping),
                        count(tone_mapping), count(duration_mapping))
  state = OrderedDict((k, _draw(seed, k, shape, sharp)) for k, shape in layout.items())
  if gate_seed is not None:
    key = "decoder.gate_layer.linear_layer.weight"
    state[key] = _draw(gate_seed, key, layout[key], sharp)
  if gate_bias is not None:
    state["decoder.gate_layer.linear_layer.bias"] = torch.tensor([gate_bias], dtype=torch.float32)

  ckpt: "OrderedDict[str, object]" = OrderedDict()
  ckpt["hparams"] = dataclasses.asdict(hp)
  ckpt["model_state"] = state
  ckpt["iteration"] = 1
  ckpt["learning_rate"] = hp.learning_rate
  ckpt["symbol_mapping"] = symbol_mapping
  if speaker_mapping is not None:
    ckpt["speaker_mapping"] = speaker_mapping
  if stress_mapping is not None:
    ckpt["stress_mapping"] = stress_mapping
  if tone_mapping is not None:
    ckpt["tone_mapping"] = tone_mapping
  if duration_mapping is not None:
    ckpt["duration_mapping"] = duration_mapping
  return ckpt


def checkpoint_dims(ckpt: Dict[str, object]) -> ModelDims:
  from tacotron_amd.config import hparams_from_dict
  hp, _ = hparams_from_dict(ckpt["hparams"])

  def count(key):
    return len(ckpt[key]) + 1 if key in ckpt else None

  return model_dims(hp, count("symbol_mapping"), count("stress_mapping"), count("speaker_mapping"),
                    count("tone_mapping"), count("duration_mapping"))


# Named configurations used by goldens, tests and the benchmark.
def tiny_hparams(**overrides) -> HParams:
  """Small extents (E = 24 + 5 one-hot = 29, H_e = 15, M = 38) so a full checkpoint is < 1 MB."""
  base = dict(symbols_embedding_dim=24, speakers_embedding_dim=8, prenet_dim=32, attention_rnn_dim=64,
              decoder_rnn_dim=64, attention_dim=16, attention_location_n_filters=4,
              attention_location_kernel_size=7, postnet_embedding_dim=32, n_mel_channels=20)
  base.update(overrides)
  return HParams(**base)


def thchs_hparams(**overrides) -> HParams:
  """Tone + duration one-hot channels, no stress channel, multi-speaker (E = 524, M = 540 with
  7 tone ids, 5 duration ids)."""
  base = dict(use_stress_embedding=False, use_tone_embedding=True, use_duration_embedding=True,
              symbols_are_ipa=True)
  base.update(overrides)
  return HParams(**base)
