"""ctypes front of oracle/libtaco_oracle.so - the CPU checker (test infrastructure only).

May be imported by tests/, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py``; never by the product package.  ``OracleModel`` mirrors the call surface of the
reference's ``Tacotron2.inference`` (src/tacotron/model.py:751-791) on host memory, with the
prenet keep-masks passed in explicitly instead of drawn from the global RNG.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from tacotron_amd import abi
from tacotron_amd.config import HParams, hparams_from_dict

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtaco_oracle.so")
_lib = None


def build(force: bool = False) -> str:
  src = os.path.join(_HERE, "taco_oracle.c")
  if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
    subprocess.check_call(["make", "-C", _HERE, "-B", "libtaco_oracle.so"], stdout=subprocess.DEVNULL)
  return _LIB_PATH


def lib():
  global _lib
  if _lib is None:
    if not os.path.exists(_LIB_PATH):
      build()
    L = C.CDLL(_LIB_PATH)
    i32p, u8p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    i64p = C.POINTER(C.c_int64)
    L.taco_oracle_threads.restype = C.c_int
    L.taco_oracle_set_threads.argtypes = [C.c_int]
    L.taco_oracle_tap_stride.argtypes = [C.POINTER(abi.TacoDims)]
    L.taco_oracle_tap_stride.restype = C.c_int32
    L.taco_oracle_encode.argtypes = [C.POINTER(abi.TacoDims), C.POINTER(abi.TacoWeights), C.POINTER(i64p),
                                     i64p, i32p, C.c_int32, C.c_int32, f32p, f32p]
    L.taco_oracle_decode.argtypes = [C.POINTER(abi.TacoDims), C.POINTER(abi.TacoWeights), f32p, f32p, i32p,
                                     C.c_int32, C.c_int32, u8p, C.c_int32, C.c_int32, C.c_float, f32p,
                                     f32p, f32p, f32p, i32p, u8p, f32p, C.c_int32]
    L.taco_oracle_postnet.argtypes = [C.POINTER(abi.TacoDims), C.POINTER(abi.TacoWeights), f32p, i32p,
                                      C.c_int32, C.c_int32, f32p]
    _lib = L
  return _lib


def _p(a: Optional[np.ndarray], ctype):
  if a is None:
    return None
  assert a.flags["C_CONTIGUOUS"]
  return a.ctypes.data_as(C.POINTER(ctype))


class OracleModel:
  def __init__(self, checkpoint: Dict[str, object], custom_hparams: Optional[Dict[str, str]] = None):
    from tacotron_amd.config import apply_custom_hparams
    hp, _ = hparams_from_dict(checkpoint["hparams"])
    self.hparams: HParams = apply_custom_hparams(hp, custom_hparams)
    counts = abi.mapping_counts(checkpoint, self.hparams)
    self.cdims, self.dims = abi.build_dims(self.hparams, **counts)
    self.cweights, self._keep = abi.bind_weights(self.hparams, self.dims, checkpoint["model_state"],
                                                 torch.device("cpu"))
    self.gate_threshold = float(self.hparams.gate_threshold)

  # ---- stage-level calls on numpy arrays ----
  def encode(self, ids: Sequence[np.ndarray], speakers: Optional[np.ndarray],
             lens: Optional[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    ids = [np.ascontiguousarray(i, dtype=np.int64) for i in ids]
    B, T = ids[0].shape
    arr = (C.POINTER(C.c_int64) * len(ids))(*[_p(i, C.c_int64) for i in ids])
    spk = None if speakers is None else np.ascontiguousarray(speakers, dtype=np.int64)
    ln = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    mem = np.zeros((B, T, self.dims.M), dtype=np.float32)
    pm = np.zeros((B, T, self.dims.A), dtype=np.float32)
    rc = lib().taco_oracle_encode(C.byref(self.cdims), C.byref(self.cweights), arr, _p(spk, C.c_int64),
                                  _p(ln, C.c_int32), B, T, _p(mem, C.c_float), _p(pm, C.c_float))
    if rc != 0:
      raise RuntimeError(f"taco_oracle_encode failed: {rc}")
    return mem, pm

  def decode(self, memory: np.ndarray, pm: np.ndarray, lens: Optional[np.ndarray], masks: np.ndarray,
             max_steps: int, forced_mel: Optional[np.ndarray] = None, T_cap: Optional[int] = None,
             want_align: bool = True, n_tap_steps: int = 0, gate_threshold: Optional[float] = None):
    B, T, _ = memory.shape
    T_cap = int(T_cap or max_steps)
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    assert masks.shape[0] >= max_steps and masks.shape[1:] == (2, self.dims.P)
    ln = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    mel = np.zeros((B, self.dims.n_mel, T_cap), dtype=np.float32)
    gate = np.zeros((B, T_cap), dtype=np.float32)
    align = np.zeros((B, T_cap, T), dtype=np.float32) if want_align else None
    n_frames = np.zeros((B,), dtype=np.int32)
    reached = np.zeros((B,), dtype=np.uint8)
    taps = None
    if n_tap_steps > 0:
      stride = lib().taco_oracle_tap_stride(C.byref(self.cdims))
      taps = np.zeros((B, n_tap_steps, stride), dtype=np.float32)
    if forced_mel is not None:
      forced_mel = np.ascontiguousarray(forced_mel, dtype=np.float32)
      assert forced_mel.shape == (B, self.dims.n_mel, T_cap)
    thr = self.gate_threshold if gate_threshold is None else gate_threshold
    rc = lib().taco_oracle_decode(
      C.byref(self.cdims), C.byref(self.cweights), _p(np.ascontiguousarray(memory), C.c_float),
      _p(np.ascontiguousarray(pm), C.c_float), _p(ln, C.c_int32), B, T, _p(masks, C.c_uint8), T_cap,
      max_steps, C.c_float(thr), _p(forced_mel, C.c_float), _p(mel, C.c_float), _p(gate, C.c_float),
      _p(align, C.c_float), _p(n_frames, C.c_int32), _p(reached, C.c_uint8), _p(taps, C.c_float),
      n_tap_steps)
    if rc != 0:
      raise RuntimeError(f"taco_oracle_decode failed: {rc}")
    return mel, gate, align, n_frames, reached, taps

  def postnet(self, mel: np.ndarray, n_frames: Optional[np.ndarray]) -> np.ndarray:
    B, _, T_cap = mel.shape
    nf = None if n_frames is None else np.ascontiguousarray(n_frames, dtype=np.int32)
    out = np.zeros_like(mel)
    rc = lib().taco_oracle_postnet(C.byref(self.cdims), C.byref(self.cweights),
                                   _p(np.ascontiguousarray(mel), C.c_float), _p(nf, C.c_int32), B, T_cap,
                                   _p(out, C.c_float))
    if rc != 0:
      raise RuntimeError(f"taco_oracle_postnet failed: {rc}")
    return out

  def split_taps(self, taps: np.ndarray) -> Dict[str, np.ndarray]:
    d = self.dims
    names = [("prenet", d.P), ("attention_hidden", d.R_a), ("attention_cell", d.R_a), ("context", d.M),
             ("decoder_hidden", d.R_d), ("decoder_cell", d.R_d)]
    out, off = {}, 0
    for name, n in names:
      out[name] = taps[..., off:off + n]
      off += n
    return out

  # ---- whole-path call, B = 1, shaped like the reference's Tacotron2.inference ----
  def inference(self, symbols, stresses, tones, durations, speakers, max_decoder_steps: int,
                masks: np.ndarray, forced_mel: Optional[np.ndarray] = None):
    ids = [np.asarray(x) for x in (symbols, stresses, tones, durations) if x is not None]
    spk = None if speakers is None else np.asarray(speakers)
    mem, pm = self.encode(ids, spk, None)
    mel, gate, align, n_frames, reached, _ = self.decode(mem, pm, None, masks, max_decoder_steps,
                                                         forced_mel=forced_mel)
    n = int(n_frames[0])
    post = self.postnet(mel, n_frames)
    return (mel[:, :, :n], post[:, :, :n], gate[:, :n, None], align[:, :n, :], bool(reached[0]))


def threads() -> int:
  return int(lib().taco_oracle_threads())
