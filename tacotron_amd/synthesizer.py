"""``Synthesizer``: checkpoint -> model on the GPU, symbol lines -> mel spectrograms.

Same constructor, attributes, ``infer`` signature and ``InferenceResult`` record as the
reference's synthesis driver (reference: src/tacotron/synthesizer.py:24-241), with the model's
arithmetic done by the HIP library (``tacotron_amd.model.Tacotron2``).  ``infer_batch`` is an
addition: the reference synthesizes one line per call.
"""
from __future__ import annotations

import logging
import os
import random
import queue
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Set, Tuple, Union

import numpy as np
import torch

from tacotron_amd import frontend
from tacotron_amd.checkpoint import (KEY_DURATION_MAPPING, KEY_MODEL_STATE, KEY_SPEAKER_MAPPING,
                                     KEY_STRESS_MAPPING, KEY_SYMBOL_MAPPING, KEY_TONE_MAPPING, CheckpointDict,
                                     get_hparams, validate)
from tacotron_amd.config import apply_custom_hparams
from tacotron_amd.model import Tacotron2


@dataclass
class InferenceResult:
  sampling_rate: int
  reached_max_decoder_steps: bool
  inference_duration_s: float
  mel_outputs_postnet: np.ndarray
  mel_outputs: Optional[np.ndarray]
  gate_outputs: Optional[np.ndarray]
  alignments: Optional[np.ndarray]
  unmappable_symbols: Optional[Set[str]]
  unmappable_stresses: Optional[Set[str]]
  unmappable_tones: Optional[Set[str]]
  unmappable_durations: Optional[Set[str]]
  duration_s: float


def get_default_device() -> torch.device:
  """First GPU (reference: src/tacotron/utils.py:29-35, which falls back to the CPU; this
  implementation has no CPU path and raises instead)."""
  if torch.cuda.device_count() < 1:
    raise RuntimeError("no GPU is visible; tacotron_amd needs an MI355X (there is no CPU path)")
  return torch.device("cuda:0")


def init_global_seeds(seed: int) -> None:
  """Seeds every generator the reference seeds before an inference call
  (reference: src/tacotron/utils.py:440-451)."""
  os.environ["PYTHONHASHSEED"] = str(seed)
  random.seed(seed)
  np.random.seed(seed)
  # torch.manual_seed(seed) = the CPU generator + every accelerator back-end torch knows; for the back-ends this
  # machine does not have it queues a deferred call together with a formatted stack trace (10 ms per call, and the
  # queue grows with every line synthesized), so the two generators that exist are seeded directly
  torch.default_generator.manual_seed(int(seed))
  if torch.cuda.is_available():
    torch.cuda.manual_seed_all(int(seed))


def spectrogram_duration_s(n_frames: int, n_fft: int, hop_length: int, sr: int = 22050) -> float:
  """What ``librosa.get_duration(S=mel, n_fft=, hop_length=)`` returns for a centred spectrogram
  with ``n_frames`` columns; the reference does not pass a sampling rate, so librosa's default
  22050 applies whatever the checkpoint's rate is (reference: src/tacotron/synthesizer.py:219-220)."""
  n_samples = n_fft + hop_length * (n_frames - 1)
  n_samples = n_samples - 2 * int(n_fft // 2)
  return float(n_samples) / sr


STAGING_SETS = 3        # batches whose host arrays may be alive at once (one being written, one being copied, one free)


@dataclass
class _Prepared:
  ids: Dict[str, Optional[List[int]]]
  speaker_id: Optional[int]
  unmappable_symbols: Optional[Set[str]]
  unmappable_stresses: Optional[Set[str]]
  unmappable_tones: Optional[Set[str]]
  unmappable_durations: Optional[Set[str]]


class Synthesizer:
  def __init__(self, checkpoint: CheckpointDict, custom_hparams: Optional[Dict[str, str]] = None,
               device: Optional[Union[str, torch.device]] = None, precision: str = "fp32"):
    """The reference's constructor (src/tacotron/synthesizer.py:47-91) plus ``precision``: "fp32" (the reference's
    arithmetic, the default) or "bf16" for the decoder of batches (see ``Tacotron2.set_precision``)."""
    validate(checkpoint)
    self.device = torch.device(device) if device is not None else get_default_device()
    hparams = apply_custom_hparams(get_hparams(checkpoint), custom_hparams)
    self.hparams = hparams

    def mapping(key: str, used: bool):
      if not used:
        return None
      if key not in checkpoint:
        raise KeyError(f"checkpoint lacks '{key}' although its hparams use that channel")
      return checkpoint[key]

    self.symbol_mapping = checkpoint[KEY_SYMBOL_MAPPING]
    self.stress_mapping = mapping(KEY_STRESS_MAPPING, hparams.use_stress_embedding)
    self.tone_mapping = mapping(KEY_TONE_MAPPING, hparams.use_tone_embedding)
    self.duration_mapping = mapping(KEY_DURATION_MAPPING, hparams.use_duration_embedding)
    self.speaker_mapping = mapping(KEY_SPEAKER_MAPPING, hparams.use_speaker_embedding)

    def count(m):
      return None if m is None else frontend.get_mappings_count(m)

    model = Tacotron2(hparams, n_symbols=count(self.symbol_mapping), n_stresses=count(self.stress_mapping),
                      n_speakers=count(self.speaker_mapping), n_tones=count(self.tone_mapping),
                      n_durations=count(self.duration_mapping))
    model.load_state_dict(checkpoint[KEY_MODEL_STATE])
    self.model = model.to(self.device).eval()
    if precision != "fp32":
      self.model.set_precision(precision)

  def get_sampling_rate(self) -> int:
    return self.hparams.sampling_rate

  def _sync(self) -> None:
    if self.device.type == "cuda":
      torch.cuda.synchronize(self.device)

  def _staging_pool(self) -> "queue.Queue":
    """Page-locked staging sets for the device-to-host copies of batches: a batch keeps its set until its arrays have
    been consumed (``finish_batch`` ... ``release``), so that the next batch can be decoded meanwhile."""
    pool = getattr(self, "_staging_sets", None)
    if pool is None:
      pool = queue.Queue()
      for _ in range(STAGING_SETS):
        pool.put({})
      self._staging_sets = pool
    return pool

  def _to_host(self, t: torch.Tensor, which: int, staging: Dict[int, torch.Tensor], wait: bool = True) -> np.ndarray:
    """Copies a device tensor into a page-locked staging buffer of the set ``staging`` (one buffer per output array; a
    pageable destination makes the runtime bounce the copy through its own small pinned buffers at a fraction of the
    link's rate) and returns the buffer as an array: valid until the set is used again.  ``wait=False`` only enqueues
    the copy on the current stream."""
    if t.device.type != "cuda":
      return t.contiguous().numpy()
    have = staging.get(which)
    if have is None or have.numel() < t.numel():
      have = torch.empty(max(t.numel(), 1), dtype=t.dtype, pin_memory=True)
      staging[which] = have
    dst = have[:t.numel()].view(t.shape)
    dst.copy_(t, non_blocking=True)
    if wait:
      self._sync()
    return dst.numpy()

  # ---- host logic: symbols -> ids, dropping positions no mapping knows ---------------------------
  def _prepare(self, symbols: Sequence[str], speaker: Optional[str]) -> _Prepared:
    logger = logging.getLogger(__name__)
    cores, stresses, tones, durations = frontend.get_map_keys(symbols, self.hparams)
    speaker_id = None
    if self.hparams.use_speaker_embedding:
      if speaker not in self.speaker_mapping:
        raise KeyError(f"Speaker '{speaker}' is not in the checkpoint's speaker mapping")
      speaker_id = self.speaker_mapping[speaker]

    unmappable: Set[int] = set()

    def channel(keys, mapping, what):
      if keys is None:
        return None, None
      ids = frontend.get_mapped_indices(keys, mapping)
      bad = {i for i, v in enumerate(ids) if v is None}
      unknown = None
      if bad:
        unknown = {keys[i] for i in bad}
        logger.warning(f"Unknown {what}: {' '.join(sorted(unknown))}")
        unmappable.update(bad)
      return ids, unknown

    stress_ids, bad_stresses = channel(stresses, self.stress_mapping, "stress(es)")
    tone_ids, bad_tones = channel(tones, self.tone_mapping, "tone(s)")
    duration_ids, bad_durations = channel(durations, self.duration_mapping, "duration(s)")
    symbol_ids, bad_symbols = channel(cores, self.symbol_mapping, "symbol(s)")

    def keep(ids):
      return None if ids is None else [v for i, v in enumerate(ids) if i not in unmappable]

    logger.debug("|".join(s if i not in unmappable else f"[{s}]" for i, s in enumerate(symbols)))
    return _Prepared(
      ids=dict(symbols=keep(symbol_ids), stresses=keep(stress_ids), tones=keep(tone_ids),
               durations=keep(duration_ids)),
      speaker_id=speaker_id, unmappable_symbols=bad_symbols, unmappable_stresses=bad_stresses,
      unmappable_tones=bad_tones, unmappable_durations=bad_durations)

  def _result(self, prep: _Prepared, mel_post: np.ndarray, reached: bool, seconds: float, mel=None, gate=None,
              align=None) -> InferenceResult:
    return InferenceResult(
      sampling_rate=self.hparams.sampling_rate, reached_max_decoder_steps=reached,
      inference_duration_s=seconds, mel_outputs_postnet=mel_post, mel_outputs=mel, gate_outputs=gate,
      alignments=align, unmappable_symbols=prep.unmappable_symbols, unmappable_stresses=prep.unmappable_stresses,
      unmappable_tones=prep.unmappable_tones, unmappable_durations=prep.unmappable_durations,
      duration_s=spectrogram_duration_s(mel_post.shape[-1], self.hparams.filter_length, self.hparams.hop_length))

  def infer(self, symbols: Sequence[str], speaker: Optional[str], max_decoder_steps: int = 3000, seed: int = 0,
            include_stats: bool = False, keep_on_device: bool = False) -> InferenceResult:
    """One line, as the reference's ``Synthesizer.infer`` (src/tacotron/synthesizer.py:93-239).  With
    ``keep_on_device`` the arrays of the result stay ``torch`` tensors on the synthesizer's device (same shapes
    and dtypes) for a consumer on the same GPU, e.g. a vocoder: nothing is copied to the host, and the tensors
    are ready on the device's current stream."""
    prep = self._prepare(symbols, speaker)
    init_global_seeds(seed)

    def tensor(ids):
      return None if ids is None else torch.tensor([ids], dtype=torch.int64, device=self.device)

    symbol_tensor = tensor(prep.ids["symbols"])
    speaker_tensor = None
    if self.hparams.use_speaker_embedding:
      speaker_tensor = torch.full(tuple(symbol_tensor.shape), prep.speaker_id, dtype=torch.int64, device=self.device)

    self._sync()
    start = time.perf_counter()
    mel, mel_post, gate, align, reached = self.model.inference(
      symbols=symbol_tensor, stresses=tensor(prep.ids["stresses"]), tones=tensor(prep.ids["tones"]),
      durations=tensor(prep.ids["durations"]), speakers=speaker_tensor, max_decoder_steps=max_decoder_steps)
    self._sync()
    seconds = time.perf_counter() - start

    def to_np(t):
      return t.squeeze(0) if keep_on_device else t.squeeze(0).cpu().numpy()

    res = self._result(prep, to_np(mel_post), reached, seconds)
    if include_stats:
      res.mel_outputs, res.gate_outputs, res.alignments = to_np(mel), to_np(gate), to_np(align)
    return res

  def prepare_batch(self, lines: Sequence[Sequence[str]], speaker: Optional[str]) -> List[_Prepared]:
    """The host-side half of ``infer_batch`` (symbols -> ids, unknown symbols dropped): pure Python, so a caller may
    run it for the next batch on another thread while the GPU decodes the current one."""
    preps = [self._prepare(line, speaker) for line in lines]
    if preps and min(len(p.ids["symbols"]) for p in preps) < 1:
      raise ValueError("a line has no mappable symbol")
    return preps

  def infer_batch(self, lines: Sequence[Sequence[str]], speaker: Optional[str], max_decoder_steps: int = 3000,
                  seed: int = 0, include_stats: bool = False, keep_on_device: bool = False,
                  prepared: Optional[List[_Prepared]] = None) -> List[InferenceResult]:
    """Synthesizes several lines in one padded batch; every line gets the result ``infer`` would
    give it (same seed for each, as the CLI reuses one seed for all lines:
    src/tacotron_cli/inference.py:137-141).  ``inference_duration_s`` is the batch time / lines.
    ``keep_on_device``: as in ``infer``; each line's arrays are views into the batch's device tensors.
    ``prepared``: what ``prepare_batch(lines, speaker)`` returned, if the caller ran it ahead of time."""
    if len(lines) == 0:
      return []
    pending = self.launch_batch(lines, speaker, max_decoder_steps, seed, include_stats, keep_on_device, prepared)
    results = self.finish_batch(pending)
    if not keep_on_device:
      for res in results:                      # the staging set goes back: the arrays must own their memory
        res.mel_outputs_postnet = res.mel_outputs_postnet.copy()
        if include_stats:
          res.mel_outputs, res.gate_outputs, res.alignments = res.mel_outputs.copy(), res.gate_outputs.copy(), res.alignments.copy()
    pending.release()
    return results

  def launch_batch(self, lines: Sequence[Sequence[str]], speaker: Optional[str], max_decoder_steps: int = 3000,
                   seed: int = 0, include_stats: bool = False, keep_on_device: bool = False,
                   prepared: Optional[List[_Prepared]] = None, tags: Optional[Sequence[object]] = None,
                   carry: Optional["_Carry"] = None, leave_unfinished: bool = False) -> "_PendingBatch":
    """The device half of ``infer_batch``: encoder, decoder (the call returns when the decoder has finished), postnet
    and the copies to the host are enqueued, the copies are NOT waited for.  ``finish_batch`` waits for them and cuts
    the batch into per-line results - a caller with many batches runs it on another thread while this thread launches
    the next batch, so the GPU does not idle while the host slices and writes.
    ``leave_unfinished`` / ``carry`` (models with ``supports_resume``): the decoder stops as soon as one of its sixteen
    slots has no successor instead of running the batch's last lines to their end with the other slots empty; the
    lines still running are in ``pending.unfinished`` and go into the next launch as its ``carry``, where they continue
    from their next frame (bit for bit as if never interrupted).  ``tags`` name the lines (default: their index);
    ``pending.finished_tags`` are those of the results ``finish_batch`` returns."""
    old_preps = list(carry.preps) if carry else []
    old_tags = list(carry.tags) if carry else []
    new_preps = list(prepared if prepared is not None else self.prepare_batch(lines, speaker))
    preps = old_preps + new_preps
    all_tags = old_tags + list(tags if tags is not None else range(len(new_preps)))
    lens = [len(p.ids["symbols"]) for p in preps]
    new_lens = lens[len(old_preps):]
    T = max(new_lens) if new_lens else 0
    init_global_seeds(seed)

    def padded(name):
      if not new_preps or new_preps[0].ids[name] is None:
        return None
      out = np.zeros((len(new_preps), T), dtype=np.int64)
      for b, p in enumerate(new_preps):
        out[b, :new_lens[b]] = p.ids[name]
      return torch.from_numpy(out).to(self.device)

    speakers = None
    if self.hparams.use_speaker_embedding and new_preps:
      speakers = torch.full((len(new_preps), T), new_preps[0].speaker_id, dtype=torch.int64, device=self.device)
    lens_t = torch.tensor(new_lens, dtype=torch.int32) if new_preps else None      # (host: the model reads them there too)
    start = time.perf_counter()
    left = None
    if carry or leave_unfinished:
      mel, post, gate, align, n_frames, reached, left = self.model.inference_batch_resumable(
        padded("symbols"), padded("stresses"), padded("tones"), padded("durations"), speakers, lens_t, max_decoder_steps,
        carry.rows if carry else None, leave_unfinished, want_align=include_stats)
    else:
      mel, post, gate, align, n_frames, reached = self.model.inference_batch(
        padded("symbols"), padded("stresses"), padded("tones"), padded("durations"), speakers, lens_t,
        max_decoder_steps, want_align=include_stats)
      # (the decoder has finished - its call waits for it -, so the frame counts are final)
      n_frames = n_frames.cpu().tolist()
      reached = reached.cpu().tolist()
    pending = _PendingBatch(self, preps, lens, n_frames, reached, include_stats, keep_on_device, start)
    pending.rows = [b for b in range(len(preps)) if reached[b] != 2]
    pending.finished_tags = [all_tags[b] for b in pending.rows]
    if left is not None:
      pending.unfinished = _Carry([all_tags[b] for b in left.rows], [preps[b] for b in left.rows], left)
    if keep_on_device:
      pending.arrays = (post, mel, gate, align)
      return pending
    # one copy per array for the whole batch (up to its longest finished utterance), sliced per line on the host
    n_max = max([n_frames[b] for b in pending.rows] or [1])
    staging = self._staging_pool().get()
    pending.staging = staging
    post_h = self._to_host(post[:, :, :n_max], 0, staging, wait=False)
    mel_h = gate_h = align_h = None
    if include_stats:
      mel_h, gate_h, align_h = (self._to_host(mel[:, :, :n_max], 1, staging, wait=False),
                                self._to_host(gate[:, :n_max], 2, staging, wait=False),
                                self._to_host(align[:, :n_max], 3, staging, wait=False))
    pending.arrays = (post_h, mel_h, gate_h, align_h)
    if self.device.type == "cuda":
      pending.copied = torch.cuda.Event()
      pending.copied.record(torch.cuda.current_stream(self.device))
    pending.keep = (post, mel, gate, align)      # the device tensors live until the copies have been waited for
    return pending

  def finish_batch(self, pending: "_PendingBatch") -> List[InferenceResult]:
    """Waits for a launched batch's copies and returns its per-line results.  Unless the batch stayed on the device,
    the arrays are views into the batch's page-locked staging set: valid until ``pending.release()``."""
    if pending.copied is not None:
      pending.copied.synchronize()
    pending.keep = None
    seconds = (time.perf_counter() - pending.start) / len(pending.preps)
    post, mel, gate, align = pending.arrays
    on_device = pending.keep_on_device
    out = []
    for b in pending.rows:                      # (rows left unfinished come back with a later batch)
      p = pending.preps[b]
      n = pending.n_frames[b]
      res = self._result(p, post[b, :, :n], bool(pending.reached[b]), seconds)
      if pending.include_stats:
        res.mel_outputs = mel[b, :, :n]
        res.gate_outputs = gate[b, :n, None] if not on_device else gate[b, :n].unsqueeze(-1)
        res.alignments = align[b, :n, :pending.lens[b]]
      out.append(res)
    return out


class _Carry:
  """The lines a launch left unfinished: their tags and prepared inputs, and what the model needs to take them up
  again in the next launch (``tacotron_amd.model.CarriedRows``, one row per line, in this order)."""

  def __init__(self, tags: List[object], preps: List[_Prepared], rows) -> None:
    self.tags, self.preps, self.rows = tags, preps, rows

  def __len__(self) -> int:
    return len(self.tags)


class _PendingBatch:
  """A batch between ``launch_batch`` and ``finish_batch`` / ``release``."""

  def __init__(self, synth, preps, lens, n_frames, reached, include_stats, keep_on_device, start):
    self.synth, self.preps, self.lens, self.n_frames, self.reached = synth, preps, lens, n_frames, reached
    self.include_stats, self.keep_on_device, self.start = include_stats, keep_on_device, start
    self.staging = None
    self.arrays = None
    self.copied = None
    self.keep = None
    self.rows: List[int] = list(range(len(preps)))      # rows whose utterances have finished
    self.finished_tags: List[object] = list(range(len(preps)))
    self.unfinished: Optional["_Carry"] = None       # lines left to the next launch (None: every line has finished)

  def release(self) -> None:
    """Gives the batch's staging set back (its result arrays must not be used afterwards)."""
    if self.staging is not None:
      self.synth._staging_pool().put(self.staging)
      self.staging = None
