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
