"""Leaving utterances unfinished and resuming them (taco_decode_resumable): a ragged set of lines decoded in several
calls, each but the last told to stop as soon as a slot has no successor, must give every line bit for bit the frames
of one uninterrupted call."""
import numpy as np
import pytest
import torch

from conftest import scenario_checkpoint
from tacotron_amd import scenarios
from tacotron_amd.masks import prenet_keep_masks
from test_gpu_parity import chans, dev, gpu_model

pytestmark = pytest.mark.gpu


def _encode(gm, ck, lens, seed):
  ids = scenarios.random_ids(ck, lens, seed=seed)
  ln = dev(np.array(lens, np.int32))
  mem, pm = gm.encode([dev(c) for c in chans(ids)], dev(ids["speakers"]), ln)
  return mem, pm, ln


def test_lines_decoded_across_calls_equal_one_uninterrupted_call(gate_table):
  gate = gate_table["ljs"]
  gm = gpu_model("ljs", gate)
  ck = scenario_checkpoint("ljs", gate)
  r = np.random.RandomState(5)
  n_lines, max_steps = 60, 200
  lens = r.randint(20, 201, size=n_lines).tolist()
  masks = prenet_keep_masks(1111, max_steps, gm.dims.P)
  mem, pm, ln = _encode(gm, ck, lens, seed=77)
  T = mem.shape[1]
  want = gm.decode(mem, pm, ln, masks, max_steps)                         # one call, everything to its end
  torch.cuda.synchronize()
  nf_want = want[3].cpu().tolist()
  assert len(set(nf_want)) > 4 and max(nf_want) == max_steps             # stops at different frames, some at the limit

  # the same lines in calls of 22, 22 and 16 new lines; what a call leaves unfinished heads the next call
  got_mel = [None] * n_lines
  got_gate = [None] * n_lines
  got_align = [None] * n_lines
  got_nf = [None] * n_lines
  carried = []                                                            # (line, slot, frames so far, mel, gate, align rows)
  calls = [list(range(0, 22)), list(range(22, 44)), list(range(44, 60))]
  n_left_total = 0
  for c, new in enumerate(calls):
    last = c == len(calls) - 1
    lines = [x[0] for x in carried] + new
    idx = torch.tensor(lines, device="cuda:0")
    resume = torch.tensor([x[1] for x in carried] + [-1] * len(new), dtype=torch.int32, device="cuda:0")
    out = gm.decode(mem[idx], pm[idx], ln[idx], masks, max_steps, resume_slots=resume if carried else None,
                    leave_unfinished=not last)
    mel, gate_o, align, nf, reached, slots = out
    for b, x in enumerate(carried):                                      # the columns the earlier calls produced
      n0 = x[2]
      mel[b, :, :n0] = x[3][:, :n0]
      gate_o[b, :n0] = x[4][:n0]
      align[b, :n0, :] = 0.0
      align[b, :n0, :x[5].shape[1]] = x[5][:n0, :]
    torch.cuda.synchronize()
    nf_h, reached_h, slots_h = nf.cpu().tolist(), reached.cpu().tolist(), slots.cpu().tolist()
    carried = []
    for b, line in enumerate(lines):
      if reached_h[b] == 2:
        assert not last and 0 <= slots_h[b] < 16 and nf_h[b] % 4 == 0 and 0 <= nf_h[b] < max_steps
        carried.append((line, slots_h[b], nf_h[b], mel[b].clone(), gate_o[b].clone(), align[b, :, :lens[line]].clone()))
      else:
        assert slots_h[b] == -1
        got_nf[line] = nf_h[b]
        got_mel[line], got_gate[line], got_align[line] = mel[b].clone(), gate_o[b].clone(), align[b, :, :lens[line]].clone()
        assert reached_h[b] == (1 if nf_h[b] == max_steps and nf_want[line] == max_steps and want[4][line].item() == 1 else 0)
    n_left_total += len(carried)
    assert len({x[1] for x in carried}) == len(carried)
  assert n_left_total > 0 and not carried
  assert got_nf == nf_want
  for line in range(n_lines):
    n = nf_want[line]
    assert torch.equal(got_mel[line][:, :n], want[0][line][:, :n]), line
    assert torch.equal(got_gate[line][:n], want[1][line][:n]), line
    assert torch.equal(got_align[line][:n, :], want[2][line][:n, :lens[line]]), line


def test_only_slots_the_last_launch_left_can_be_resumed(gate_table):
  """A resume request must name a slot in which the handle's LAST lockstep launch left an utterance: a slot nobody was
  left in, a slot named twice, a slot number out of range, and a slot of the launch before the last are refused with
  the library's argument error instead of continuing from whatever the saved state holds; the handle decodes on."""
  from tacotron_amd._lib import TacoError, TacoTimeoutError
  gate = gate_table["ljs"]
  gm = gpu_model("ljs", gate)
  ck = scenario_checkpoint("ljs", gate)
  lens = np.random.RandomState(8).randint(20, 201, size=24).tolist()
  max_steps = 120
  masks = prenet_keep_masks(1111, max_steps, gm.dims.P)
  mem, pm, ln = _encode(gm, ck, lens, seed=5)
  minus = torch.full((24,), -1, dtype=torch.int32, device="cuda:0")

  def refused(resume):
    with pytest.raises(TacoError) as err:
      gm.decode(mem, pm, ln, masks, max_steps, resume_slots=resume, leave_unfinished=False)
    assert not isinstance(err.value, TacoTimeoutError) and "resume_slot" in str(err.value)

  gm.decode(mem, pm, ln, masks, max_steps)                       # a plain call leaves nothing
  one = minus.clone(); one[0] = 3
  refused(one)
  out = gm.decode(mem, pm, ln, masks, max_steps, leave_unfinished=True)
  torch.cuda.synchronize()
  left = [(b, s) for b, (r, s) in enumerate(zip(out[4].cpu().tolist(), out[5].cpu().tolist())) if r == 2]
  assert left, "the calibrated stop leaves some of 24 lines running when the first slot frees"
  free = sorted(set(range(16)) - {s for _, s in left})
  if free:
    bad = minus.clone(); bad[left[0][0]] = free[0]
    refused(bad)                                                 # nobody was left in that slot
  twice = minus.clone(); twice[left[0][0]] = left[0][1]; twice[(left[0][0] + 1) % 24] = left[0][1]
  refused(twice)
  far = minus.clone(); far[left[0][0]] = 16
  refused(far)
  # the refusals above enqueued nothing: the left utterances can still be taken up, and give the one-call frames
  want = gm.decode(mem, pm, ln, masks, max_steps)
  gm.decode(mem, pm, ln, masks, max_steps, leave_unfinished=True)
  torch.cuda.synchronize()
  good = minus.clone()
  rows = [b for b, _ in left]
  out2 = gm.decode(mem, pm, ln, masks, max_steps, leave_unfinished=True)
  left2 = [(b, s) for b, (r, s) in enumerate(zip(out2[4].cpu().tolist(), out2[5].cpu().tolist())) if r == 2]
  assert [b for b, _ in left2] == rows                           # same lines, same launch
  idx = torch.tensor(rows, device="cuda:0")
  resume = torch.tensor([s for _, s in left2], dtype=torch.int32, device="cuda:0")
  fin = gm.decode(mem[idx].contiguous(), pm[idx].contiguous(), ln[idx].contiguous(), masks, max_steps, resume_slots=resume)
  for j, b in enumerate(rows):
    n0, n = int(out2[3][b]), int(want[3][b])
    assert int(fin[3][j]) == n
    assert torch.equal(fin[0][j][:, n0:n], want[0][b][:, n0:n])
  stale = torch.tensor([s for _, s in left2], dtype=torch.int32, device="cuda:0")     # ... and not a second time
  with pytest.raises(TacoError):
    gm.decode(mem[idx].contiguous(), pm[idx].contiguous(), ln[idx].contiguous(), masks, max_steps, resume_slots=stale)
