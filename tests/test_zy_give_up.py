"""The give-up path of the persistent decoders.  Their 256 workgroups wait for each other's values, so all of them must
be running at once; when something else holds a compute unit during the call, the waits run into their bound, the launch
drains and the call reports TACO_ETIMEDOUT (a TimeoutError in Python) instead of hanging - and the handle works again
as soon as the GPU is free."""
import ctypes as C
import os
import time

import pytest
import torch

from conftest import golden_ids, load_golden
from tacotron_amd.masks import prenet_keep_masks
from test_gpu_parity import chans, dev, gpu_model

pytestmark = pytest.mark.gpu

GIVE_UP_BOUND_S = 2.0         # a launch gives up after 2^18 polls of a wait: a few tenths of a second


def _hold_a_cu(stream, seconds):
  """One small workgroup that keeps 64 KiB of one CU's LDS for `seconds` on `stream` (tests/support/hold_cu.hip, built
  by __graft_entry__.build()): a decoder workgroup - which needs nearly all of a CU's LDS - cannot start there."""
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "support", "libhold_cu.so")
  lib = C.CDLL(path)
  lib.hold_cu.argtypes = [C.c_void_p, C.c_double]
  assert lib.hold_cu(C.c_void_p(stream.cuda_stream), float(seconds)) == 0


@pytest.mark.parametrize("batch", [1, 3])
def test_decode_gives_up_when_a_cu_is_held_and_works_again_afterwards(batch, gate_table):
  gm = gpu_model("ljs", gate_table["ljs"])
  z, _ = load_golden("ljs")
  ids = golden_ids(z)
  rep = lambda a: None if a is None else dev(a).repeat(batch, 1)   # noqa: E731
  mem, pm = gm.encode([rep(c) for c in chans(ids)], rep(ids["speakers"]), None)
  masks = prenet_keep_masks(3, 40, gm.dims.P)
  want = gm.decode(mem, pm, None, masks, 40, gate_threshold=2.0)
  torch.cuda.synchronize()
  assert want[3].cpu().tolist() == [40] * batch

  side = torch.cuda.Stream(device="cuda:0")
  _hold_a_cu(side, 2.5)
  time.sleep(0.05)                                     # the holder is running before the decoder is enqueued
  t0 = time.perf_counter()
  with pytest.raises(TimeoutError) as err:
    gm.decode(mem, pm, None, masks, 40, gate_threshold=2.0)
  waited = time.perf_counter() - t0
  assert "gave up" in str(err.value) or "wait" in str(err.value), str(err.value)
  assert waited < GIVE_UP_BOUND_S, waited
  print(f"gave up after {waited:.2f} s: {err.value}")
  torch.cuda.synchronize()                             # the held CU is free again

  got = gm.decode(mem, pm, None, masks, 40, gate_threshold=2.0)
  torch.cuda.synchronize()
  for a, b in zip(got, want):
    assert torch.equal(a, b)
