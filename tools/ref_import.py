"""Makes the reference's model modules importable in the build container.

Only for tools/ (golden generation, reference timing).  The reference's package ``__init__``
eagerly imports plotting/metric modules whose third-party packages are absent here; none of
them touches the model arithmetic, so they are registered as empty stand-in modules.  Nothing
under tests/, bench.py or the product imports this file, and /root/reference does not exist
on the GPU box.
"""
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def _stub(name, **attrs):
  mod = types.ModuleType(name)
  mod.__dict__.update(attrs)
  sys.modules[name] = mod
  return mod


def install():
  if "tacotron" in sys.modules:
    return
  fdtw = _stub("fastdtw")
  sub = _stub("fastdtw.fastdtw", fastdtw=None)
  fdtw.fastdtw = sub

  def get_duration(S=None, n_fft=2048, hop_length=512, sr=22050, **_):
    # librosa.get_duration for a spectrogram input: (frames - 1) * hop / sr, sr defaulting to 22050
    n_frames = S.shape[-1]
    n_samples = n_fft + hop_length * (n_frames - 1)
    n_samples = n_samples - 2 * int(n_fft // 2)
    return float(n_samples) / sr

  _stub("librosa", get_duration=get_duration)
  _stub("librosa.filters", mel=None)
  _stub("librosa.util", normalize=None, pad_center=None, tiny=None)
  _stub("skimage")
  _stub("skimage.metrics", structural_similarity=None)
  sys.path.insert(0, REFERENCE_SRC)
