#!/usr/bin/env python3
"""Times the encoder stage of the benchmark workload (input vectors, convolutions, BiLSTM, memory tail)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import abi, scenarios
from tacotron_amd.checkpoint import get_hparams
from tacotron_amd.model import Tacotron2
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ckpt = scenarios.checkpoint("ljs")
hp = get_hparams(ckpt)
m = Tacotron2(hp, **abi.mapping_counts(ckpt, hp)); m.load_state_dict(ckpt["model_state"]); m.to("cuda:0")
sym, st, spk = bench.build_inputs(ckpt)
d = lambda a: torch.from_numpy(a).to("cuda:0")
ids, sp = [d(sym), d(st)], d(spk)
for _ in range(3):
  m.encode(ids, sp, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
  m.encode(ids, sp, None)
e1.record()
torch.cuda.synchronize()
print(f"encode T_in={sym.shape[1]}: {e0.elapsed_time(e1) / reps * 1000:.1f} us per call [{os.environ.get('TACO_HIP_LIB', 'product')}]")
