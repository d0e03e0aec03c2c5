"""The entry point that allocates nothing, taco_postnet_noalloc (scratch inside the library's code object), against
taco_postnet through the C ABI: identical output (same kernels, same inputs: bit for bit) on two calls in a row, the
second with a smaller input, and a clear refusal beyond the built-in scratch."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import gpu_model

pytestmark = pytest.mark.gpu


def _postnet(gm, fn, mel, nf):
  from tacotron_amd import _lib
  out = torch.full_like(mel, float("nan"))
  B, _, T_cap = mel.shape
  _lib.check(fn(gm._require(), C.c_void_p(mel.data_ptr()), C.c_void_p(nf.data_ptr()), B, T_cap,
                C.c_void_p(out.data_ptr()), gm._stream()))
  torch.cuda.synchronize()
  return out


def test_postnet_noalloc_equals_postnet_on_two_calls_in_a_row():
  from tacotron_amd import _lib
  z, meta = load_golden("ljs")
  gm = gpu_model(meta["scenario"], meta["gate"])
  L = _lib.lib()
  g = torch.Generator().manual_seed(7)
  for B, T_cap, frames in ((3, 600, [600, 417, 1]), (2, 90, [37, 90])):          # the second call is smaller
    mel = torch.randn(B, gm.dims.n_mel, T_cap, generator=g).to("cuda:0")
    nf = torch.tensor(frames, dtype=torch.int32, device="cuda:0")
    for b, n in enumerate(frames):
      mel[b, :, n:] = 0.0
    want = _postnet(gm, L.taco_postnet, mel, nf)
    got = _postnet(gm, L.taco_postnet_noalloc, mel, nf)
    assert bool(torch.isfinite(got).all())
    assert torch.equal(got, want)
    for b, n in enumerate(frames):
      assert not got[b, :, n:].any()
  # beyond the built-in scratch: refused with a message, nothing enqueued
  cols = 2048
  too_big = torch.zeros(2, gm.dims.n_mel, cols, device="cuda:0")
  nf = torch.tensor([cols, cols], dtype=torch.int32, device="cuda:0")
  with pytest.raises(_lib.TacoError) as err:
    _postnet(gm, L.taco_postnet_noalloc, too_big, nf)
  assert "scratch" in str(err.value)


def test_postnet_noalloc_may_overlap_the_encoder_of_the_same_handle():
  """taco_postnet_noalloc keeps off every workspace of the handle (activations, partial convolution tiles and their
  counters are all inside the built-in scratch), so it may run on one stream while taco_encode of the same handle runs
  on another: both results equal those of the same calls made one after the other, bit for bit, every time."""
  from tacotron_amd import _lib
  from conftest import golden_ids
  from test_gpu_parity import chans, dev
  z, meta = load_golden("ljs")
  gm = gpu_model(meta["scenario"], meta["gate"])
  L = _lib.lib()
  ids = golden_ids(z)
  enc_in = ([dev(c) for c in chans(ids)], dev(ids["speakers"]), None)
  g = torch.Generator().manual_seed(11)
  B, T_cap = 1, 250                      # 250 frames: every layer's tiles are shared by several workgroups (partial tiles in use)
  mel = torch.randn(B, gm.dims.n_mel, T_cap, generator=g).to("cuda:0")
  nf = torch.tensor([T_cap], dtype=torch.int32, device="cuda:0")
  mem_want, pm_want = gm.encode(*enc_in)
  post_want = _postnet(gm, L.taco_postnet_noalloc, mel, nf)
  torch.cuda.synchronize()
  side = torch.cuda.Stream(device="cuda:0")
  for _ in range(20):
    out = torch.full_like(mel, float("nan"))
    with torch.cuda.stream(side):
      _lib.check(L.taco_postnet_noalloc(gm._require(), C.c_void_p(mel.data_ptr()), C.c_void_p(nf.data_ptr()), B, T_cap,
                                        C.c_void_p(out.data_ptr()), C.c_void_p(side.cuda_stream)))
    mem, pm = gm.encode(*enc_in)
    torch.cuda.synchronize()
    assert torch.equal(out, post_want)
    assert torch.equal(mem, mem_want) and torch.equal(pm, pm_want)
