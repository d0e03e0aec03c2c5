"""ctypes mirror of ``include/taco.h`` and the checkpoint -> struct binding.

``TacoDims`` / ``TacoWeights`` are laid out exactly as ``taco_dims`` / ``taco_weights`` in the C
header.  ``bind_weights`` fills a ``TacoWeights`` with the data pointers of a checkpoint's
``model_state`` tensors (state-dict key names as the reference saves them:
src/tacotron/model.py:612-688) after moving them to one device as contiguous fp32.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from tacotron_amd.config import HParams, ModelDims, channel_widths, model_dims

MAX_CHANNELS = 4
MAX_CONVS = 8

TACO_OK = 0
TACO_EINVAL = -22
TACO_ENOMEM = -12
TACO_EHIP = -5
TACO_ETIMEDOUT = -110

_f32p = C.POINTER(C.c_float)


class TacoDims(C.Structure):
  _fields_ = [(n, C.c_int32) for n in (
    "E", "H_e", "S", "M", "P", "R_a", "R_d", "A", "F", "K", "n_mel", "C_post", "k_post", "n_post",
    "k_enc", "n_enc", "n_channels")] + [
    ("ch_width", C.c_int32 * MAX_CHANNELS),
    ("ch_rows", C.c_int32 * MAX_CHANNELS),
    ("spk_rows", C.c_int32),
  ]


class TacoWeights(C.Structure):
  _fields_ = [
    ("ch_table", _f32p * MAX_CHANNELS),
    ("spk_table", _f32p),
    ("enc_conv_w", _f32p * MAX_CONVS), ("enc_conv_b", _f32p * MAX_CONVS),
    ("enc_bn_w", _f32p * MAX_CONVS), ("enc_bn_b", _f32p * MAX_CONVS),
    ("enc_bn_mean", _f32p * MAX_CONVS), ("enc_bn_var", _f32p * MAX_CONVS),
    ("enc_lstm_w_ih", _f32p * 2), ("enc_lstm_w_hh", _f32p * 2),
    ("enc_lstm_b_ih", _f32p * 2), ("enc_lstm_b_hh", _f32p * 2),
    ("prenet_w1", _f32p), ("prenet_w2", _f32p),
    ("att_w_ih", _f32p), ("att_w_hh", _f32p), ("att_b_ih", _f32p), ("att_b_hh", _f32p),
    ("query_w", _f32p), ("memory_w", _f32p), ("v_w", _f32p),
    ("loc_conv_w", _f32p), ("loc_dense_w", _f32p),
    ("dec_w_ih", _f32p), ("dec_w_hh", _f32p), ("dec_b_ih", _f32p), ("dec_b_hh", _f32p),
    ("proj_w", _f32p), ("proj_b", _f32p), ("gate_w", _f32p), ("gate_b", _f32p),
    ("post_conv_w", _f32p * MAX_CONVS), ("post_conv_b", _f32p * MAX_CONVS),
    ("post_bn_w", _f32p * MAX_CONVS), ("post_bn_b", _f32p * MAX_CONVS),
    ("post_bn_mean", _f32p * MAX_CONVS), ("post_bn_var", _f32p * MAX_CONVS),
  ]


def build_dims(hp: HParams, n_symbols: int, n_stresses: Optional[int], n_speakers: Optional[int],
               n_tones: Optional[int], n_durations: Optional[int]) -> Tuple[TacoDims, ModelDims]:
  md = model_dims(hp, n_symbols, n_stresses, n_speakers, n_tones, n_durations)
  if md.n_enc > MAX_CONVS or md.n_post > MAX_CONVS:
    raise ValueError("too many convolution layers")
  widths = channel_widths(hp, n_symbols, n_stresses, n_tones, n_durations)
  rows = [n_symbols]
  if hp.use_stress_embedding:
    rows.append(n_stresses)
  if hp.use_tone_embedding:
    rows.append(n_tones)
  if hp.use_duration_embedding:
    rows.append(n_durations)
  d = TacoDims()
  for name in ("E", "H_e", "S", "M", "P", "R_a", "R_d", "A", "F", "K", "n_mel", "C_post", "k_post",
               "n_post", "k_enc", "n_enc"):
    setattr(d, name, getattr(md, name))
  d.n_channels = len(widths)
  for i, (wd, r) in enumerate(zip(widths, rows)):
    d.ch_width[i] = int(wd)
    d.ch_rows[i] = int(r)
  d.spk_rows = int(n_speakers) if hp.use_speaker_embedding else 0
  return d, md


def _ptr(t: torch.Tensor):
  return C.cast(C.c_void_p(t.data_ptr()), _f32p)


def bind_weights(hp: HParams, md: ModelDims, state: Dict[str, torch.Tensor],
                 device: torch.device, cdims: Optional[TacoDims] = None) -> Tuple[TacoWeights, List[torch.Tensor]]:
  """Returns the filled struct and the list of tensors whose storage it points into (keep the
  list alive for as long as the pointers are used).  With ``cdims`` (the extents handed to the
  library) the embedding and speaker tables are checked against the row counts and widths the library
  will copy, as the reference's ``load_state_dict`` would refuse a table of another shape."""
  keep: List[torch.Tensor] = []

  def get(key: str, shape: Tuple[int, ...]):
    if key not in state:
      raise KeyError(f"checkpoint model_state lacks '{key}'")
    t = state[key].detach().to(device=device, dtype=torch.float32).contiguous()
    if tuple(t.shape) != tuple(shape):
      raise ValueError(f"'{key}' has shape {tuple(t.shape)}, expected {tuple(shape)}")
    keep.append(t)
    return _ptr(t)

  w = TacoWeights()
  # input channels, in concatenation order symbol, stress, tone, duration
  chans = [("symbol_embeddings.weight", hp.train_symbol_with_embedding, True)]
  chans.append(("stress_embedding.weight", hp.train_stress_with_embedding, hp.use_stress_embedding))
  chans.append(("tone_embedding.weight", hp.train_tone_with_embedding, hp.use_tone_embedding))
  chans.append(("duration_embedding.weight", hp.train_duration_with_embedding, hp.use_duration_embedding))
  slot = 0
  for key, is_table, used in chans:
    if not used:
      continue
    if is_table:
      if key not in state:
        raise KeyError(f"checkpoint model_state lacks '{key}'")
      expect = tuple(state[key].shape) if cdims is None else (int(cdims.ch_rows[slot]), int(cdims.ch_width[slot]))
      w.ch_table[slot] = get(key, expect)
    slot += 1
  if hp.use_speaker_embedding and hp.train_speaker_with_embedding:
    if "speakers_embeddings.weight" not in state:
      raise KeyError("checkpoint model_state lacks 'speakers_embeddings.weight'")
    rows = state["speakers_embeddings.weight"].shape[0] if cdims is None else int(cdims.spk_rows)
    w.spk_table = get("speakers_embeddings.weight", (rows, md.S))

  def conv_stack(prefix: str, n: int, cin0: int, cmid: int, cout_last: int, k: int, dst: str):
    for i in range(n):
      c_in = cin0 if i == 0 else cmid
      c_out = cout_last if i == n - 1 else cmid
      getattr(w, f"{dst}_conv_w")[i] = get(f"{prefix}.{i}.0.conv.weight", (c_out, c_in, k))
      getattr(w, f"{dst}_conv_b")[i] = get(f"{prefix}.{i}.0.conv.bias", (c_out,))
      getattr(w, f"{dst}_bn_w")[i] = get(f"{prefix}.{i}.1.weight", (c_out,))
      getattr(w, f"{dst}_bn_b")[i] = get(f"{prefix}.{i}.1.bias", (c_out,))
      getattr(w, f"{dst}_bn_mean")[i] = get(f"{prefix}.{i}.1.running_mean", (c_out,))
      getattr(w, f"{dst}_bn_var")[i] = get(f"{prefix}.{i}.1.running_var", (c_out,))

  conv_stack("encoder.convolutions", md.n_enc, md.E, md.E, md.E, md.k_enc, "enc")
  for i, suffix in enumerate(("", "_reverse")):
    w.enc_lstm_w_ih[i] = get(f"encoder.lstm.weight_ih_l0{suffix}", (4 * md.H_e, md.E))
    w.enc_lstm_w_hh[i] = get(f"encoder.lstm.weight_hh_l0{suffix}", (4 * md.H_e, md.H_e))
    w.enc_lstm_b_ih[i] = get(f"encoder.lstm.bias_ih_l0{suffix}", (4 * md.H_e,))
    w.enc_lstm_b_hh[i] = get(f"encoder.lstm.bias_hh_l0{suffix}", (4 * md.H_e,))

  dec = "decoder."
  att = dec + "attention_layer."
  w.prenet_w1 = get(dec + "prenet.layers.0.linear_layer.weight", (md.P, md.n_mel))
  w.prenet_w2 = get(dec + "prenet.layers.1.linear_layer.weight", (md.P, md.P))
  w.att_w_ih = get(dec + "attention_rnn.weight_ih", (4 * md.R_a, md.P + md.M))
  w.att_w_hh = get(dec + "attention_rnn.weight_hh", (4 * md.R_a, md.R_a))
  w.att_b_ih = get(dec + "attention_rnn.bias_ih", (4 * md.R_a,))
  w.att_b_hh = get(dec + "attention_rnn.bias_hh", (4 * md.R_a,))
  w.query_w = get(att + "query_layer.linear_layer.weight", (md.A, md.R_a))
  w.memory_w = get(att + "memory_layer.linear_layer.weight", (md.A, md.M))
  w.v_w = get(att + "v.linear_layer.weight", (1, md.A))
  w.loc_conv_w = get(att + "location_layer.location_conv.conv.weight", (md.F, 2, md.K))
  w.loc_dense_w = get(att + "location_layer.location_dense.linear_layer.weight", (md.A, md.F))
  w.dec_w_ih = get(dec + "decoder_rnn.weight_ih", (4 * md.R_d, md.R_a + md.M))
  w.dec_w_hh = get(dec + "decoder_rnn.weight_hh", (4 * md.R_d, md.R_d))
  w.dec_b_ih = get(dec + "decoder_rnn.bias_ih", (4 * md.R_d,))
  w.dec_b_hh = get(dec + "decoder_rnn.bias_hh", (4 * md.R_d,))
  w.proj_w = get(dec + "linear_projection.linear_layer.weight", (md.n_mel, md.R_d + md.M))
  w.proj_b = get(dec + "linear_projection.linear_layer.bias", (md.n_mel,))
  w.gate_w = get(dec + "gate_layer.linear_layer.weight", (1, md.R_d + md.M))
  w.gate_b = get(dec + "gate_layer.linear_layer.bias", (1,))
  conv_stack("postnet.convolutions", md.n_post, md.n_mel, md.C_post, md.n_mel, md.k_post, "post")
  return w, keep


def mapping_counts(ckpt: Dict[str, object], hp: HParams) -> Dict[str, Optional[int]]:
  """Table sizes = mapping entries + 1 for the padding id (reference:
  src/tacotron/frontend/main.py:29-30, src/tacotron/synthesizer.py:78-86)."""
  def count(key: str, used: bool) -> Optional[int]:
    if not used:
      return None
    if key not in ckpt:
      raise KeyError(f"checkpoint lacks '{key}'")
    return len(ckpt[key]) + 1
  return dict(
    n_symbols=count("symbol_mapping", True),
    n_stresses=count("stress_mapping", hp.use_stress_embedding),
    n_speakers=count("speaker_mapping", hp.use_speaker_embedding),
    n_tones=count("tone_mapping", hp.use_tone_embedding),
    n_durations=count("duration_mapping", hp.use_duration_embedding),
  )
