"""``Tacotron2`` with the reference's call surface, computed by the HIP library.

Keeps the constructor, ``load_state_dict`` / ``to`` / ``eval`` and the
``inference(symbols, stresses, tones, durations, speakers, max_decoder_steps)`` method of the
reference's module (reference: src/tacotron/model.py:612-688, 751-791) so that code written
against it - in particular ``Synthesizer`` - runs unchanged, but it holds no torch arithmetic:
every stage is a call into ``libtaco_hip.so`` on the device's current stream.  torch is used for
device memory, streams and the CPU random generator that defines the prenet dropout masks.
"""
from __future__ import annotations

import ctypes as C
import hashlib
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from tacotron_amd import _lib, abi
from tacotron_amd.config import HParams

_UNLIMITED_START = 8192     # first capacity tried when max_decoder_steps == 0 (no limit)


class CarriedRows:
  """Utterances a resumable decode left unfinished, as the next call takes them: per utterance the slot that holds its
  state, its frames so far and its length, and - stacked, one row per utterance - the encoder's output for it and what
  has been written of its mel / gate / alignment rows.  ``rows`` are their indices in the call that left them."""

  def __init__(self, slots: List[int], n_frames: List[int], lengths: List[int], rows: List[int], memory: torch.Tensor,
               processed_memory: torch.Tensor, mel: torch.Tensor, gate: torch.Tensor, align: Optional[torch.Tensor]):
    self.slots, self.n_frames, self.lengths, self.rows = slots, n_frames, lengths, rows
    self.memory, self.processed_memory = memory, processed_memory        # [n, T_in of that call, M] / [n, T_in, A]
    self.mel, self.gate, self.align = mel, gate, align                  # [n, n_mel, T_cap], [n, T_cap], [n, T_cap, T_in] or None


class Tacotron2:
  def __init__(self, hparams: HParams, n_symbols: int, n_stresses: Optional[int], n_speakers: Optional[int],
               n_tones: Optional[int], n_durations: Optional[int]):
    self.hparams = hparams
    self.counts = dict(n_symbols=n_symbols, n_stresses=n_stresses, n_speakers=n_speakers, n_tones=n_tones,
                       n_durations=n_durations)
    self.cdims, self.dims = abi.build_dims(hparams, **self.counts)
    self.gate_threshold = float(hparams.gate_threshold)
    self.n_mel_channels = hparams.n_mel_channels
    self._state: Optional[Dict[str, torch.Tensor]] = None
    self._handle: Optional[C.c_void_p] = None
    self._keep: List[torch.Tensor] = []
    self.device: Optional[torch.device] = None
    self._mask_cache: Dict[bytes, torch.Tensor] = {}
    self.training = False

  # ---- module-like plumbing -------------------------------------------------------------------
  def load_state_dict(self, state: Dict[str, torch.Tensor]) -> None:
    self._state = state
    if self._handle is not None:
      self._rebuild()

  def eval(self) -> "Tacotron2":
    return self

  def to(self, device, non_blocking: bool = False) -> "Tacotron2":
    device = torch.device(device)
    if device.type != "cuda":
      raise RuntimeError(f"this Tacotron2 runs on an AMD GPU only (got device '{device}'); there is no CPU path")
    if device.index is None:
      device = torch.device("cuda", torch.cuda.current_device())
    self.device = device
    self._rebuild()
    return self

  def _rebuild(self) -> None:
    if self._state is None:
      raise RuntimeError("load_state_dict must be called before moving the model to a device")
    self.close()
    weights, keep = abi.bind_weights(self.hparams, self.dims, self._state, self.device, self.cdims)
    handle = C.c_void_p()
    torch.cuda.synchronize(self.device)
    _lib.check(_lib.lib().taco_create(C.byref(self.cdims), C.byref(weights), self.device.index, C.byref(handle)))
    self._handle = handle
    del keep          # the library owns a copy now

  def close(self) -> None:
    if self._handle is not None:
      _lib.lib().taco_destroy(self._handle)
      self._handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  @property
  def step_weight_bytes(self) -> int:
    return int(_lib.lib().taco_step_weight_bytes(self._require()))

  def _require(self) -> C.c_void_p:
    if self._handle is None:
      raise RuntimeError("model is not on a device yet: call .to('cuda:N') first")
    return self._handle

  def _stream(self) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  # ---- prenet dropout masks -------------------------------------------------------------------
  def masks_from_global_rng(self, n_steps: int) -> torch.Tensor:
    """uint8 [n_steps, 2, P] on the device, drawn from torch's global CPU generator exactly as the
    reference's prenet would draw them during this call (see tacotron_amd.masks).  The draw for a
    given generator state is cached, so a caller that re-seeds with the same seed before every
    line (as the CLI does) uploads the masks once."""
    state = torch.get_rng_state()
    key = hashlib.sha1(state.numpy().tobytes()).digest() + n_steps.to_bytes(4, "little")
    hit = self._mask_cache.get(key)
    if hit is not None:
      return hit
    from tacotron_amd import masks as masks_mod
    P = self.dims.P
    masks_mod.prenet_keep_masks(0, 1, P)   # settles whether one bulk draw reproduces the per-call stream
    if masks_mod._bulk_equals_stepwise[P]:
      flat = torch.empty(n_steps * 2 * P, dtype=torch.float32).bernoulli_(0.5)
      host = flat.to(torch.uint8).view(n_steps, 2, P)
    else:
      host = torch.empty(n_steps, 2, P, dtype=torch.uint8)
      row = torch.empty(1, P, dtype=torch.float32)
      for t in range(n_steps):
        for layer in range(2):
          row.bernoulli_(0.5)
          host[t, layer] = row[0].to(torch.uint8)
    dev = host.to(self.device)
    if len(self._mask_cache) > 8:
      self._mask_cache.clear()
    self._mask_cache[key] = dev
    return dev

  # ---- stages ------------------------------------------------------------------------------------
  def encode(self, ids: Sequence[torch.Tensor], speakers: Optional[torch.Tensor],
             lens: Optional[torch.Tensor], validate: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Enqueues the encoder stage and returns at once.  Ids outside their tables are detected on the
    device and raise an IndexError at the next synchronising call (``decode``), or right here with
    ``validate=True`` (which waits for the stage to finish)."""
    h = self._require()
    ids = [i.to(device=self.device, dtype=torch.int64).contiguous() for i in ids]
    if len(ids) != self.cdims.n_channels:
      raise ValueError(f"expected {self.cdims.n_channels} id tensors, got {len(ids)}")
    B, T = ids[0].shape
    for c, t in enumerate(ids):
      if tuple(t.shape) != (B, T):
        raise ValueError("id tensors differ in shape")
    spk = None
    if self.dims.S > 0:
      if speakers is None:
        raise ValueError("the checkpoint models speakers: a speakers tensor is required")
      spk = speakers.to(device=self.device, dtype=torch.int64).contiguous()
    ln = None if lens is None else lens.to(device=self.device, dtype=torch.int32).contiguous()
    memory = torch.empty(B, T, self.dims.M, device=self.device, dtype=torch.float32)
    pm = torch.empty(B, T, self.dims.A, device=self.device, dtype=torch.float32)
    arr = (C.c_void_p * len(ids))(*[C.c_void_p(t.data_ptr()) for t in ids])
    _lib.check(_lib.lib().taco_encode(
      h, arr, None if spk is None else C.c_void_p(spk.data_ptr()),
      None if ln is None else C.c_void_p(ln.data_ptr()), B, T, C.c_void_p(memory.data_ptr()),
      C.c_void_p(pm.data_ptr()), self._stream()))
    if validate:
      _lib.check(_lib.lib().taco_encode_status(h, self._stream()))
    return memory, pm

  def decode(self, memory: torch.Tensor, pm: torch.Tensor, lens: Optional[torch.Tensor], masks: torch.Tensor,
             max_steps: int, forced_mel: Optional[torch.Tensor] = None, T_cap: Optional[int] = None,
             want_align: bool = True, gate_threshold: Optional[float] = None,
             resume_slots: Optional[torch.Tensor] = None, leave_unfinished: bool = False):
    """The decoder stage.  Returns ``(mel, gate, align, n_frames, reached)``; with ``resume_slots`` or
    ``leave_unfinished`` (``taco_decode_resumable``: a call may end while utterances are still running and a later call
    takes them up again, see include/taco.h) a sixth element, ``slots`` [B] int32: -1, or the slot an utterance was left
    in (its ``reached`` is 2 and its ``n_frames`` the frames it has so far)."""
    h = self._require()
    B, T, _ = memory.shape
    T_cap = int(T_cap or max_steps)
    masks = masks.to(device=self.device, dtype=torch.uint8).contiguous()
    if masks.shape[0] < max_steps or tuple(masks.shape[1:]) != (2, self.dims.P):
      raise ValueError(f"masks must be [>= {max_steps}, 2, {self.dims.P}], got {tuple(masks.shape)}")
    ln = None if lens is None else lens.to(device=self.device, dtype=torch.int32).contiguous()
    mel = torch.empty(B, self.dims.n_mel, T_cap, device=self.device, dtype=torch.float32)
    gate = torch.empty(B, T_cap, device=self.device, dtype=torch.float32)
    align = torch.empty(B, T_cap, T, device=self.device, dtype=torch.float32) if want_align else None
    n_frames = torch.zeros(B, device=self.device, dtype=torch.int32)
    reached = torch.zeros(B, device=self.device, dtype=torch.uint8)
    if forced_mel is not None:
      forced_mel = forced_mel.to(device=self.device, dtype=torch.float32).contiguous()
      if tuple(forced_mel.shape) != (B, self.dims.n_mel, T_cap):
        raise ValueError("forced_mel must be [B, n_mel, T_cap]")
    thr = self.gate_threshold if gate_threshold is None else float(gate_threshold)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    if resume_slots is None and not leave_unfinished:
      _lib.check(_lib.lib().taco_decode(
        h, p(memory.contiguous()), p(pm.contiguous()), p(ln), B, T, p(masks), T_cap, int(max_steps), C.c_float(thr),
        p(forced_mel), p(mel), p(gate), p(align), p(n_frames), p(reached), self._stream()))
      return mel, gate, align, n_frames, reached
    rs = None if resume_slots is None else resume_slots.to(device=self.device, dtype=torch.int32).contiguous()
    if rs is not None and tuple(rs.shape) != (B,):
      raise ValueError("resume_slots must be [B]")
    slots = torch.empty(B, device=self.device, dtype=torch.int32)
    _lib.check(_lib.lib().taco_decode_resumable(
      h, p(memory.contiguous()), p(pm.contiguous()), p(ln), B, T, p(masks), T_cap, int(max_steps), C.c_float(thr),
      p(forced_mel), p(mel), p(gate), p(align), p(n_frames), p(reached), p(rs), 1 if leave_unfinished else 0, p(slots),
      self._stream()))
    return mel, gate, align, n_frames, reached, slots

  def postnet(self, mel: torch.Tensor, n_frames: Optional[torch.Tensor]) -> torch.Tensor:
    h = self._require()
    B, _, T_cap = mel.shape
    out = torch.empty_like(mel)
    nf = None if n_frames is None else n_frames.to(device=self.device, dtype=torch.int32).contiguous()
    _lib.check(_lib.lib().taco_postnet(h, C.c_void_p(mel.data_ptr()), None if nf is None else C.c_void_p(nf.data_ptr()),
                                       B, T_cap, C.c_void_p(out.data_ptr()), self._stream()))
    return out

  def last_decode_timing(self) -> Tuple[float, int]:
    us, frames = C.c_double(), C.c_int32()
    _lib.check(_lib.lib().taco_last_decode_timing(self._require(), C.byref(us), C.byref(frames)))
    return us.value, frames.value

  def record_frame_clock(self, capacity: int) -> None:
    """From now on utterance 0 of every decode leaves one reading of the device's 100 MHz clock per frame
    (capacity frames at most; 0 switches it off).  See frame_latencies_us()."""
    if capacity > 0:
      self._frame_clock = torch.zeros(capacity, dtype=torch.int64, device=self.device)
      ptr, cap = C.c_void_p(self._frame_clock.data_ptr()), capacity
    else:
      self._frame_clock, ptr, cap = None, C.c_void_p(0), 0
    _lib.check(_lib.lib().taco_set_frame_clock(self._require(), ptr, cap))

  def set_precision(self, precision: str) -> None:
    """"fp32" (default: every result within the fp32 parity bounds) or "bf16": the decoder of batches keeps the gate
    rows of its two LSTM cells as bf16 and multiplies them with bf16-rounded activations (fp32 accumulation and
    cell states).  Faster; frames within 2e-3 of the reference's under teacher forcing, stop frames may differ.
    The reference has no such switch (it computes in fp32); nothing selects "bf16" unless the caller does."""
    codes = {"fp32": 0, "bf16": 1}
    if precision not in codes:
      raise ValueError(f"precision must be one of {sorted(codes)}, not {precision!r}")
    _lib.check(_lib.lib().taco_set_precision(self._require(), codes[precision]))
    self.precision = precision

  def frame_latencies_us(self, n_frames: int) -> torch.Tensor:
    """Latency of frames 1 .. n_frames-1 of the last decode's utterance 0 in microseconds (differences of
    consecutive clock readings, 10 ns each), on the host."""
    if getattr(self, "_frame_clock", None) is None:
      raise RuntimeError("record_frame_clock() has not been called")
    ticks = self._frame_clock[:min(n_frames, self._frame_clock.numel())].cpu()
    return (ticks[1:] - ticks[:-1]).to(torch.float64) * 0.01

  # ---- whole path ------------------------------------------------------------------------------
  supports_resume = True     # decode() can leave utterances unfinished and take them up again (taco_decode_resumable)

  def inference_batch_resumable(self, symbols: Optional[torch.Tensor], stresses: Optional[torch.Tensor], tones: Optional[torch.Tensor],
                                durations: Optional[torch.Tensor], speakers: Optional[torch.Tensor], lens: Optional[torch.Tensor],
                                max_decoder_steps: int, carried: Optional["CarriedRows"], leave_unfinished: bool,
                                want_align: bool = True):
    """``inference_batch`` for a caller that decodes many batches in a row and does not want each to end with a tail of
    mostly empty decoder slots: with ``leave_unfinished`` the decoder stops as soon as a slot has no successor; the
    utterances still running come back with ``reached == 2``, their frames so far, and - as ``carried_out`` - what the
    next call needs to take them up again (``CarriedRows``: their slots, frame counts, lengths, and their rows of the
    encoder's output and of mel / gate / alignments; the encoder is not run again for them).  Passed as the next call's ``carried`` they are the first rows
    of that call's results, go on from their next frame - bit for bit as if never interrupted - with their earlier
    columns in place; the id tensors (``symbols`` ... ``lens``) hold the NEW utterances only (``symbols`` None: there are
    none).  The postnet skips the rows left unfinished.  ``max_decoder_steps`` must be positive.
    Returns ``(mel, mel_postnet, gate, alignments, n_frames, reached, carried_out)`` with ``n_frames`` / ``reached``
    as lists (the call has waited for the decoder anyway) and ``carried_out`` None when every utterance has finished."""
    if int(max_decoder_steps) <= 0:
      raise ValueError("leaving utterances unfinished needs a step limit")
    n_old = 0 if carried is None else len(carried.slots)
    mem_new = pm_new = None
    if symbols is not None:
      ids = [t for t in (symbols, stresses, tones, durations) if t is not None]
      mem_new, pm_new = self.encode(ids, speakers, lens)
    new_lens = [] if lens is None or symbols is None else [int(v) for v in lens.tolist()]
    if symbols is not None and lens is None:
      new_lens = [symbols.shape[1]] * symbols.shape[0]
    if n_old:
      n_new = 0 if mem_new is None else mem_new.shape[0]
      w_old = carried.memory.shape[1]
      width = max(w_old, mem_new.shape[1] if n_new else 0)
      memory = torch.zeros(n_old + n_new, width, self.dims.M, device=self.device, dtype=torch.float32)
      pm = torch.zeros(n_old + n_new, width, self.dims.A, device=self.device, dtype=torch.float32)
      memory[:n_old, :w_old] = carried.memory
      pm[:n_old, :w_old] = carried.processed_memory
      if n_new:
        memory[n_old:, :mem_new.shape[1]] = mem_new
        pm[n_old:, :pm_new.shape[1]] = pm_new
      all_lens = list(carried.lengths) + new_lens
    else:
      memory, pm, all_lens = mem_new, pm_new, new_lens
    B = memory.shape[0]
    cap = int(max_decoder_steps)
    mk = self.masks_from_global_rng(cap)
    resume = None
    if n_old:
      resume = torch.tensor(list(carried.slots) + [-1] * (B - n_old), dtype=torch.int32)
    lens_dev = torch.tensor(all_lens, dtype=torch.int32, device=self.device)
    mel, gate, align, n_frames, reached, slots = self.decode(memory, pm, lens_dev, mk, cap, want_align=want_align,
                                                            resume_slots=resume, leave_unfinished=leave_unfinished)
    if n_old:       # what earlier calls produced of these rows: one masked copy per array (this call wrote columns n0..)
      n0 = torch.tensor(carried.n_frames, dtype=torch.int32, device=self.device)
      earlier = torch.arange(cap, device=self.device, dtype=torch.int32)[None, :] < n0[:, None]
      mel[:n_old] = torch.where(earlier[:, None, :], carried.mel, mel[:n_old])
      gate[:n_old] = torch.where(earlier, carried.gate, gate[:n_old])
      if align is not None and carried.align is not None:
        w = min(align.shape[2], carried.align.shape[2])
        align[:n_old, :, :w] = torch.where(earlier[:, :, None], carried.align[:, :, :w], align[:n_old, :, :w])
    done = torch.where(reached == 2, torch.zeros_like(n_frames), n_frames)
    post = self.postnet(mel, done)
    # one copy to the host for the three small per-utterance arrays (the decoder has finished: its call waited for it)
    host = torch.stack([n_frames, reached.to(torch.int32), slots]).cpu().tolist()
    n_frames_l, reached_l, slots_l = host
    left = [b for b in range(B) if reached_l[b] == 2]
    carried_out = None
    if left:
      idx = torch.tensor(left, dtype=torch.int64, device=self.device)
      carried_out = CarriedRows(slots=[slots_l[b] for b in left], n_frames=[n_frames_l[b] for b in left],
                                lengths=[all_lens[b] for b in left], rows=left,
                                memory=memory.index_select(0, idx), processed_memory=pm.index_select(0, idx),
                                mel=mel.index_select(0, idx), gate=gate.index_select(0, idx),
                                align=None if align is None else align.index_select(0, idx))
    return mel, post, gate, align, n_frames_l, reached_l, carried_out

  def inference_batch(self, symbols: torch.Tensor, stresses: Optional[torch.Tensor], tones: Optional[torch.Tensor],
                      durations: Optional[torch.Tensor], speakers: Optional[torch.Tensor],
                      lens: Optional[torch.Tensor], max_decoder_steps: int, masks: Optional[torch.Tensor] = None,
                      want_align: bool = True):
    """Padded batch [B, T_in] with per-utterance lengths; every utterance is computed exactly as an
    unpadded B=1 call would be (the reference has no batched inference).  Returns padded tensors
    ``(mel [B,n_mel,T_cap], mel_postnet, gate [B,T_cap], alignments [B,T_cap,T_in] or None,
    n_frames [B] int32, reached_max [B] uint8)``."""
    ids = [t for t in (symbols, stresses, tones, durations) if t is not None]
    memory, pm = self.encode(ids, speakers, lens)
    limit = int(max_decoder_steps)
    cap = limit if limit > 0 else _UNLIMITED_START
    rng_at_entry = torch.get_rng_state()
    while True:
      torch.set_rng_state(rng_at_entry)     # every attempt draws the same mask stream
      mk = masks if masks is not None else self.masks_from_global_rng(cap)
      mel, gate, align, n_frames, reached = self.decode(memory, pm, lens, mk, cap, want_align=want_align)
      if limit > 0 or not bool(reached.any()):
        break
      cap *= 2                  # no limit requested: rerun with more room (same seed -> same frames)
    if limit <= 0:
      reached = torch.zeros_like(reached)
    post = self.postnet(mel, n_frames)
    return mel, post, gate, align, n_frames, reached

  def inference(self, symbols: torch.Tensor, stresses: Optional[torch.Tensor], tones: Optional[torch.Tensor],
                durations: Optional[torch.Tensor], speakers: Optional[torch.Tensor], max_decoder_steps: int):
    """Same arguments, result tuple, shapes and dtypes as the reference's ``Tacotron2.inference``
    for its only supported batch size, 1: ``(mel [1,n_mel,T], mel_postnet [1,n_mel,T],
    gate [1,T,1], alignments [1,T,T_in], reached_max_decoder_steps: bool)``."""
    if symbols.dim() != 2 or symbols.shape[0] != 1:
      raise ValueError("inference() takes one utterance, shaped [1, T_in] (use inference_batch for more)")
    mel, post, gate, align, n_frames, reached = self.inference_batch(
      symbols, stresses, tones, durations, speakers, None, max_decoder_steps)
    n = int(n_frames[0])
    return (mel[:, :, :n], post[:, :, :n], gate[:, :n].unsqueeze(-1), align[:, :n, :], bool(reached[0]))
