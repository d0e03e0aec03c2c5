"""Hyper-parameter record stored in (and read back from) a checkpoint's ``hparams`` entry.

The checkpoint format keeps ``dataclasses.asdict(hparams)`` under the key ``hparams``
(reference: src/tacotron/checkpoint_handling.py:38-41, src/tacotron/hparams.py:7-120,
src/tacotron/taco_stft.py:37-51).  The mel-inference path only *computes* with the model
block, but the whole record is kept so that a checkpoint written here round-trips through
the reference's ``get_hparams`` and the other way round.  Unknown keys found in a
checkpoint are dropped with a warning (reference: checkpoint_handling.py:151-159); unknown
keys in a user override are an error (reference: src/tacotron/utils.py:577-619).
"""
from __future__ import annotations

import dataclasses
import logging
from dataclasses import dataclass, field
from math import ceil
from typing import Any, Dict, List, Optional, Set, Tuple


@dataclass
class HParams:
  # ---- run control (not used by inference, kept for format round-trips) ----
  epochs: Optional[int] = 500
  iterations: Optional[int] = 0
  iters_per_checkpoint: int = 1000
  epochs_per_checkpoint: int = 1
  seed: int = 1234
  cudnn_enabled: bool = True
  cudnn_benchmark: bool = False
  save_first_iteration: bool = True
  ignore_layers: List[str] = field(default_factory=list)
  use_saved_mels: bool = False
  cache_mels: bool = False

  # ---- spectrogram geometry ----
  filter_length: int = 1024
  hop_length: int = 256
  win_length: int = 1024
  window: str = "hann"
  n_mel_channels: int = 80
  sampling_rate: int = 22050
  mel_fmin: float = 0.0
  mel_fmax: float = 8000.0

  # ---- input channels: each is an embedding table or a one-hot block ----
  train_symbol_with_embedding: bool = True
  symbols_embedding_dim: Optional[int] = 512
  use_speaker_embedding: bool = True
  train_speaker_with_embedding: bool = True
  speakers_embedding_dim: Optional[int] = 16
  use_stress_embedding: bool = True
  train_stress_with_embedding: bool = False
  stress_embedding_dim: Optional[int] = None
  use_tone_embedding: bool = False
  train_tone_with_embedding: bool = False
  tone_embedding_dim: Optional[int] = None
  use_duration_embedding: bool = False
  train_duration_with_embedding: bool = False
  duration_embedding_dim: Optional[int] = None
  symbols_are_ipa: bool = True

  # ---- encoder ----
  encoder_kernel_size: int = 5
  encoder_n_convolutions: int = 3

  # ---- decoder ----
  n_frames_per_step: int = 1
  decoder_rnn_dim: int = 1024
  prenet_dim: int = 256
  gate_threshold: float = 0.5
  p_attention_dropout: float = 0.1
  p_decoder_dropout: float = 0.1
  attention_rnn_dim: int = 1024
  attention_dim: int = 128
  attention_location_n_filters: int = 32
  attention_location_kernel_size: int = 31

  # ---- postnet ----
  postnet_embedding_dim: int = 512
  postnet_kernel_size: int = 5
  postnet_n_convolutions: int = 5

  # ---- optimiser (not used by inference) ----
  learning_rate: float = 1e-03
  grad_clip_thresh: float = 1.0
  batch_size: int = 64
  mask_padding: bool = True
  beta1: float = 0.9
  beta2: float = 0.999
  eps: float = 1e-08
  weight_decay: float = 1e-06
  amsgrad: bool = False
  use_exponential_lr_decay: bool = False
  lr_decay_start_after_epoch: Optional[int] = 250
  lr_decay_gamma: Optional[float] = 0.97
  lr_decay_min: Optional[float] = 1e-05


def hparams_from_dict(stored: Dict[str, Any]) -> Tuple[HParams, Set[str]]:
  """Builds the record from a checkpoint's dict; returns it and the keys that were dropped."""
  known = {f.name for f in dataclasses.fields(HParams)}
  kept = {k: v for k, v in stored.items() if k in known}
  dropped = {k for k in stored if k not in known}
  if dropped:
    logging.getLogger(__name__).warning(
      "Ignored these hparams from checkpoint because they did not exist in the current HParams: "
      f"{', '.join(sorted(dropped))} (#{len(dropped)})")
  return HParams(**kept), dropped


def apply_custom_hparams(hparams: HParams, custom: Optional[Dict[str, str]]) -> HParams:
  """``--custom-hparams k=v,...`` semantics: each string is converted with the type of the
  value it replaces, the empty string means ``None``, and an unknown key raises."""
  if custom is None:
    return hparams
  current = dataclasses.asdict(hparams)
  converted: Dict[str, Any] = {}
  for key, text in custom.items():
    if key not in current:
      raise Exception(f"Unknown hparam: {key}")
    converted[key] = None if text == "" else type(current[key])(text)
  return dataclasses.replace(hparams, **converted)


def parse_hparams_string(text: Optional[str]) -> Optional[Dict[str, str]]:
  if text is None:
    return None
  return dict(part.split("=") for part in text.split(","))


@dataclass(frozen=True)
class ModelDims:
  """Tensor extents of one checkpoint, derived the way the model constructor derives them
  (reference: src/tacotron/model.py:220-260, 293-356)."""
  E: int        # encoder channel count = concatenated input-channel widths
  H_e: int      # BiLSTM hidden size per direction = ceil(E / 2)
  S: int        # speaker vector width (0 when speakers are not modelled)
  M: int        # memory width = 2*H_e + S
  P: int        # prenet width
  R_a: int      # attention LSTM width
  R_d: int      # decoder LSTM width
  A: int        # attention space width
  F: int        # location filters
  K: int        # location kernel size
  n_mel: int
  C_post: int
  k_post: int
  n_post: int
  k_enc: int
  n_enc: int

  @property
  def L_a(self) -> int:
    return self.P + self.M + self.R_a

  @property
  def L_d(self) -> int:
    return self.R_a + self.M + self.R_d

  def step_params(self) -> int:
    """Number of weights one decoder frame touches (prenet, both LSTM cells with their
    two bias vectors, query/location/v layers, mel and gate projections)."""
    d = self
    return (d.P * d.n_mel + d.P * d.P
            + 4 * d.R_a * (d.P + d.M + d.R_a + 2)
            + d.A * d.R_a + 2 * d.F * d.K + d.A * d.F + d.A
            + 4 * d.R_d * (d.R_a + d.M + d.R_d + 2)
            + (d.n_mel + 1) * (d.R_d + d.M + 1))


def channel_widths(hp: HParams, n_symbols: int, n_stresses: Optional[int], n_tones: Optional[int],
                   n_durations: Optional[int]) -> List[int]:
  widths = [hp.symbols_embedding_dim if hp.train_symbol_with_embedding else n_symbols]
  if hp.use_stress_embedding:
    widths.append(hp.stress_embedding_dim if hp.train_stress_with_embedding else n_stresses)
  if hp.use_tone_embedding:
    widths.append(hp.tone_embedding_dim if hp.train_tone_with_embedding else n_tones)
  if hp.use_duration_embedding:
    widths.append(hp.duration_embedding_dim if hp.train_duration_with_embedding else n_durations)
  return [int(w) for w in widths]


def model_dims(hp: HParams, n_symbols: int, n_stresses: Optional[int], n_speakers: Optional[int],
               n_tones: Optional[int], n_durations: Optional[int]) -> ModelDims:
  if hp.n_frames_per_step != 1:
    raise ValueError("n_frames_per_step must be 1")
  E = sum(channel_widths(hp, n_symbols, n_stresses, n_tones, n_durations))
  H_e = ceil(E / 2)
  S = 0
  if hp.use_speaker_embedding:
    S = int(hp.speakers_embedding_dim if hp.train_speaker_with_embedding else n_speakers)
  return ModelDims(
    E=E, H_e=H_e, S=S, M=2 * H_e + S, P=hp.prenet_dim, R_a=hp.attention_rnn_dim,
    R_d=hp.decoder_rnn_dim, A=hp.attention_dim, F=hp.attention_location_n_filters,
    K=hp.attention_location_kernel_size, n_mel=hp.n_mel_channels,
    C_post=hp.postnet_embedding_dim, k_post=hp.postnet_kernel_size,
    n_post=hp.postnet_n_convolutions, k_enc=hp.encoder_kernel_size,
    n_enc=hp.encoder_n_convolutions)
