"""Loader and ctypes prototypes for ``libtaco_hip.so`` (the C ABI declared in ``include/taco.h``).

There is no CPU or PyTorch fallback: if the shared library is missing or a call fails, an
exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

from tacotron_amd import abi

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# TACO_HIP_LIB selects another build of the same sources (e.g. the stage-stamp diagnostic build).
LIB_PATH = os.environ.get("TACO_HIP_LIB") or os.path.join(_PKG_DIR, "libtaco_hip.so")
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")

EXPORTED_SYMBOLS = (
  "taco_last_error", "taco_version", "taco_create", "taco_destroy", "taco_step_weight_bytes",
  "taco_encode", "taco_encode_status", "taco_decode", "taco_decode_resumable", "taco_postnet", "taco_postnet_noalloc", "taco_last_decode_timing",
  "taco_set_frame_clock", "taco_set_precision",
)

_lib: Optional[C.CDLL] = None


class TacoError(RuntimeError):
  def __init__(self, code: int, message: str):
    super().__init__(f"libtaco_hip error {code}: {message}")
    self.code = code


def build(force: bool = False, extra: str = "") -> str:
  """Compiles the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
  args = ["make", "-C", CSRC_DIR, "-j4"]
  if force:
    args.append("-B")
  if extra:
    args.append(f"EXTRA={extra}")
  subprocess.check_call(args, stdout=subprocess.DEVNULL)
  return LIB_PATH


def lib() -> C.CDLL:
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise FileNotFoundError(
      f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
      f"or `make -C {CSRC_DIR}`; there is no fallback path")
  L = C.CDLL(LIB_PATH)
  vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
  L.taco_last_error.restype = C.c_char_p
  L.taco_version.restype = C.c_char_p
  L.taco_create.argtypes = [C.POINTER(abi.TacoDims), C.POINTER(abi.TacoWeights), C.c_int, C.POINTER(vp)]
  L.taco_destroy.argtypes = [vp]
  L.taco_step_weight_bytes.argtypes = [vp]
  L.taco_step_weight_bytes.restype = C.c_int64
  L.taco_encode.argtypes = [vp, C.POINTER(vp), vp, vp, i32, i32, vp, vp, vp]
  L.taco_encode_status.argtypes = [vp, vp]
  L.taco_decode.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]
  L.taco_decode_resumable.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
  L.taco_postnet.argtypes = [vp, vp, vp, i32, i32, vp, vp]
  L.taco_postnet_noalloc.argtypes = [vp, vp, vp, i32, i32, vp, vp]
  L.taco_last_decode_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i32)]
  L.taco_set_frame_clock.argtypes = [vp, vp, i32]
  L.taco_set_precision.argtypes = [vp, i32]
  _lib = L
  return L


ERANGE = -34
ETIMEDOUT = -110


class TacoIndexError(TacoError, IndexError):
  """An input id outside its table (what the reference's embedding lookup raises an IndexError for)."""


class TacoTimeoutError(TacoError, TimeoutError):
  """A persistent kernel gave up waiting for a hand-off (TACO_ETIMEDOUT): its workgroups were not all running at
  once, which happens when something else holds compute units of the GPU during the call."""


_ERROR_TYPES = {ERANGE: TacoIndexError, ETIMEDOUT: TacoTimeoutError}


def check(rc: int) -> None:
  if rc != 0:
    message = lib().taco_last_error().decode("utf-8", "replace")
    raise _ERROR_TYPES.get(rc, TacoError)(rc, message)
