"""Test-only stand-ins that put the CPU oracle behind the product's host-side classes, so that the
host logic above the C ABI (Synthesizer, the synthesize command, the sharding over ranks) can be
exercised on machines without a GPU.  Nothing outside tests/ imports this."""
from __future__ import annotations

import numpy as np
import torch

from oracle.oracle import OracleModel
from tacotron_amd.synthesizer import Synthesizer


class OracleTacotron2:
  """``inference`` / ``inference_batch`` with the call surface of ``tacotron_amd.model.Tacotron2`` on
  CPU tensors; the prenet keep-masks are drawn from torch's global CPU generator, two [P] Bernoulli
  draws per decoder step, as the reference's always-on dropout does (src/tacotron/model.py:136-141)."""

  def __init__(self, checkpoint, custom_hparams=None):
    self.om = OracleModel(checkpoint, custom_hparams)
    self.dims = self.om.dims

  def _masks(self, n):
    return torch.empty(n, 2, self.dims.P).bernoulli_(0.5).to(torch.uint8).numpy()

  def inference_batch(self, symbols, stresses, tones, durations, speakers, lens, max_decoder_steps, masks=None,
                      want_align=True):
    ids = [t.numpy() for t in (symbols, stresses, tones, durations) if t is not None]
    spk = None if speakers is None else speakers.numpy()
    ln = None if lens is None else lens.numpy().astype(np.int32)
    mem, pm = self.om.encode(ids, spk, ln)
    mk = self._masks(max_decoder_steps) if masks is None else masks
    mel, gate, align, n_frames, reached, _ = self.om.decode(mem, pm, ln, mk, max_decoder_steps, want_align=True)
    post = self.om.postnet(mel, n_frames)
    t = torch.from_numpy
    return t(mel), t(post), t(gate), t(align), t(n_frames), t(reached)

  def inference(self, symbols, stresses, tones, durations, speakers, max_decoder_steps):
    mel, post, gate, align, n_frames, reached = self.inference_batch(symbols, stresses, tones, durations, speakers,
                                                                     None, max_decoder_steps)
    n = int(n_frames[0])
    return mel[:, :, :n], post[:, :, :n], gate[:, :n].unsqueeze(-1), align[:, :n, :], bool(reached[0])


def oracle_synthesizer(checkpoint, custom_hparams=None, device=None) -> Synthesizer:
  """A ``Synthesizer`` whose mappings and host logic are the product's and whose model is the oracle."""
  from tacotron_amd.checkpoint import get_hparams
  from tacotron_amd.config import apply_custom_hparams
  s = object.__new__(Synthesizer)
  s.device = torch.device("cpu")
  s.hparams = apply_custom_hparams(get_hparams(checkpoint), custom_hparams)
  s.symbol_mapping = checkpoint["symbol_mapping"]
  s.stress_mapping = checkpoint.get("stress_mapping") if s.hparams.use_stress_embedding else None
  s.tone_mapping = checkpoint.get("tone_mapping") if s.hparams.use_tone_embedding else None
  s.duration_mapping = checkpoint.get("duration_mapping") if s.hparams.use_duration_embedding else None
  s.speaker_mapping = checkpoint.get("speaker_mapping") if s.hparams.use_speaker_embedding else None
  s.model = OracleTacotron2(checkpoint, custom_hparams)
  return s


class OracleTacotron2Resumable(OracleTacotron2):
  """The oracle behind the resumable call surface (``supports_resume`` / ``inference_batch_resumable`` of
  ``tacotron_amd.model.Tacotron2``), for the host logic that carries unfinished lines from one launch to the next.
  The oracle keeps no decoder state, so a line is computed in full when it first appears and parked; what a real launch
  would do is only imitated at the interface: told to leave lines unfinished, the call reports the last (up to fifteen)
  of its lines that have more than eight frames to show as left, eight more frames of each than before, and hands the
  rest out when the line comes back."""
  supports_resume = True

  def __init__(self, checkpoint, custom_hparams=None):
    super().__init__(checkpoint, custom_hparams)
    self._parked = {}
    self._next_slot = 0
    self.left_total = 0

  def inference_batch_resumable(self, symbols, stresses, tones, durations, speakers, lens, max_decoder_steps, carried,
                                leave_unfinished, want_align=True):
    from tacotron_amd.model import CarriedRows
    rows = []
    for slot in (carried.slots if carried is not None else []):
      rows.append(self._parked.pop(slot))
    if symbols is not None:
      mel, post, gate, align, n_frames, reached = self.inference_batch(symbols, stresses, tones, durations, speakers, lens,
                                                                       max_decoder_steps)
      for b in range(mel.shape[0]):
        rows.append(dict(mel=mel[b], post=post[b], gate=gate[b], align=align[b], n=int(n_frames[b]), reached=int(reached[b]),
                         shown=0, length=int(lens[b]) if lens is not None else symbols.shape[1]))
    B, cap = len(rows), int(max_decoder_steps)
    width = max(r["align"].shape[1] for r in rows)
    out_mel = torch.zeros(B, self.dims.n_mel, cap)
    out_post = torch.zeros(B, self.dims.n_mel, cap)
    out_gate = torch.zeros(B, cap)
    out_align = torch.zeros(B, cap, width)
    left = [b for b in range(B) if rows[b]["n"] > rows[b]["shown"] + 8][-15:] if leave_unfinished else []
    n_out, reached_out, slots = [], [], []
    for b, r in enumerate(rows):
      if b in left:
        r["shown"] += 8
        n, code = r["shown"], 2
        slot = self._next_slot % 16
        self._next_slot += 1
        assert slot not in self._parked
        self._parked[slot] = r
        slots.append(slot)
      else:
        n, code = r["n"], r["reached"]
        out_post[b, :, :n] = r["post"][:, :n]
      out_mel[b, :, :n] = r["mel"][:, :n]
      out_gate[b, :n] = r["gate"][:n]
      out_align[b, :n, :r["align"].shape[1]] = r["align"][:n]
      n_out.append(n)
      reached_out.append(code)
    self.left_total += len(left)
    carried_out = None
    if left:
      carried_out = CarriedRows(slots=slots, n_frames=[rows[b]["shown"] for b in left], lengths=[rows[b]["length"] for b in left],
                                rows=left, memory=torch.zeros(len(left), 1, 1), processed_memory=torch.zeros(len(left), 1, 1),
                                mel=out_mel[left], gate=out_gate[left], align=out_align[left])
    return out_mel, out_post, out_gate, out_align, n_out, reached_out, carried_out


def oracle_synthesizer_resumable(checkpoint, custom_hparams=None, device=None) -> Synthesizer:
  s = oracle_synthesizer(checkpoint, custom_hparams, device)
  s.model = OracleTacotron2Resumable(checkpoint, custom_hparams)
  return s
