"""The encoder's and the postnet's output for a line must not depend on the batch it is computed in: the convolutions
add a tile's input-channel units in one fixed order whether a launch is large (one workgroup per tile) or small (one
workgroup per unit), so a line alone, in a small batch and in a large one gets the same bits - like the reference,
which has no batches at all (src/tacotron/model.py:751-791)."""
import numpy as np
import pytest
import torch

from conftest import scenario_checkpoint
from tacotron_amd import scenarios
from test_gpu_parity import chans, dev, gpu_model

pytestmark = pytest.mark.gpu


def _encode(gm, ids, rows, lens):
  sel = {k: (None if v is None else v[rows][:, :max(lens)]) for k, v in ids.items() if k != "lens"}
  mem, pm = gm.encode([dev(c) for c in chans(sel)], dev(sel["speakers"]), dev(np.array(lens, np.int32)))
  torch.cuda.synchronize()
  return mem, pm


@pytest.mark.parametrize("name", ["ljs", "thchs"])
def test_encoder_output_is_the_same_alone_and_in_batches(name, gate_table):
  gm = gpu_model(name, gate_table[name])
  ck = scenario_checkpoint(name, gate_table[name])
  r = np.random.RandomState(11)
  lens = r.randint(3, 201, size=64).tolist()
  lens[5], lens[17], lens[40] = 102, 1, 200
  ids = scenarios.random_ids(ck, lens, seed=3)
  mem64, pm64 = _encode(gm, ids, list(range(64)), lens)
  for rows in ([5], [17], [40], [5, 17], list(range(30, 47)), list(range(0, 64, 3))):
    mem, pm = _encode(gm, ids, rows, [lens[b] for b in rows])
    for j, b in enumerate(rows):
      n = lens[b]
      assert torch.equal(mem[j, :n], mem64[b, :n]), (name, rows, b)
      assert torch.equal(pm[j, :n], pm64[b, :n]), (name, rows, b)


def test_postnet_output_is_the_same_alone_and_in_batches(gate_table):
  gm = gpu_model("ljs", gate_table["ljs"])
  g = torch.Generator().manual_seed(9)
  r = np.random.RandomState(12)
  n_frames = r.randint(1, 1001, size=48).tolist()
  n_frames[0], n_frames[1], n_frames[2], n_frames[3] = 1000, 250, 64, 65
  mel = torch.randn(48, gm.dims.n_mel, 1000, generator=g)
  for b, n in enumerate(n_frames):
    mel[b, :, n:] = 0.0
  mel_d = dev(mel.numpy())
  nf = dev(np.array(n_frames, np.int32))
  post48 = gm.postnet(mel_d, nf)
  torch.cuda.synchronize()
  for rows in ([0], [1], [2], [3], [1, 2], list(range(4, 20))):
    idx = torch.tensor(rows, device=mel_d.device)
    cap = max(n_frames[b] for b in rows)                     # another capacity, another number of tiles per launch
    post = gm.postnet(mel_d[idx][:, :, :cap].contiguous(), nf[idx])
    torch.cuda.synchronize()
    for j, b in enumerate(rows):
      n = n_frames[b]
      assert torch.equal(post[j, :, :n], post48[b, :, :n]), (rows, b)
      assert not post[j, :, n:].any()
