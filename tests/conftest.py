import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
  """(arrays, meta) of tests/golden/<name>.npz - outputs of the reference on a synthetic checkpoint."""
  z = np.load(os.path.join(GOLDEN, name + ".npz"))
  meta = json.loads(bytes(z["meta_json"]).decode("utf-8"))
  return z, meta


def golden_ids(z, row=None):
  """Input id arrays of a golden; row=b extracts utterance b of a padded batch, unpadded."""
  out = {}
  for k in ("symbols", "stresses", "tones", "durations", "speakers"):
    a = z["in_" + k] if ("in_" + k) in z else None
    if a is not None and row is not None:
      a = a[row:row + 1, :int(z["in_lens"][row])]
    out[k] = a
  return out


@pytest.fixture(scope="session")
def gate_table():
  with open(os.path.join(GOLDEN, "gate_bias.json")) as f:
    return json.load(f)


_ckpt_cache = {}


def scenario_checkpoint(name, gate):
  from tacotron_amd import scenarios
  key = (name, None if gate is None else (int(gate[0]), float(gate[1])))
  if key not in _ckpt_cache:
    _ckpt_cache[key] = scenarios.checkpoint(name, gate)
  return _ckpt_cache[key]


_oracle_cache = {}


def oracle_model(name, gate):
  """CPU checker for a scenario (test infrastructure, see oracle/taco_oracle.c)."""
  from oracle.oracle import OracleModel, build
  build()
  key = (name, None if gate is None else (int(gate[0]), float(gate[1])))
  if key not in _oracle_cache:
    _oracle_cache[key] = OracleModel(scenario_checkpoint(name, gate))
  return _oracle_cache[key]
