"""TextGrid reader (both Praat text formats) and the synthesize-grids command with the CPU oracle as
the model.  The reference reads grids with the third-party ``textgrid`` package, which is not available
here: the expected values below follow Praat's file-format description, parity with that package is
unpinned."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from tacotron_amd import cli, textgrid_io

LONG = '''File type = "ooTextFile"
Object class = "TextGrid"

xmin = 0
xmax = 1.5
tiers? <exists>
size = 2
item []:
    item [1]:
        class = "IntervalTier"
        name = "words"
        xmin = 0
        xmax = 1.5
        intervals: size = 2
        intervals [1]:
            xmin = 0
            xmax = 0.7
            text = "say ""hi"""
        intervals [2]:
            xmin = 0.7
            xmax = 1.5
            text = ""
    item [2]:
        class = "IntervalTier"
        name = "symbols"
        xmin = 0
        xmax = 1.5
        intervals: size = 4
        intervals [1]:
            xmin = 0
            xmax = 0.25
            text = "s1"
        intervals [2]:
            xmin = 0.25
            xmax = 0.5
            text = "s12"
        intervals [3]:
            xmin = 0.5
            xmax = 1e0
            text = "s3"
        intervals [4]:
            xmin = 1.0
            xmax = 1.5
            text = "s30"
'''

SHORT = '''File type = "ooTextFile"
Object class = "TextGrid"

0
1.5
<exists>
2
"TextTier"
"tones"
0
1.5
1
0.4
"H*"
"IntervalTier"
"symbols"
0
1.5
3
0
0.5
"s7"
0.5
1
"s8"
1
1.5
"s9"
'''


def test_long_and_short_formats():
  tiers = textgrid_io.parse_textgrid(LONG)
  assert [t.name for t in tiers] == ["words", "symbols"]
  assert tiers[0].marks == ('say "hi"', "")
  assert textgrid_io.first_tier(tiers, "symbols").marks == ("s1", "s12", "s3", "s30")
  assert tiers[1].entries[2] == (0.5, 1.0, "s3")
  tiers = textgrid_io.parse_textgrid(SHORT)
  assert [(t.tier_class, t.name) for t in tiers] == [("TextTier", "tones"), ("IntervalTier", "symbols")]
  assert tiers[0].entries == [(0.4, 0.4, "H*")]
  assert tiers[1].marks == ("s7", "s8", "s9")
  assert textgrid_io.first_tier(tiers, "absent") is None
  with pytest.raises(textgrid_io.TextGridError):
    textgrid_io.parse_textgrid("hello")
  with pytest.raises(textgrid_io.TextGridError):
    textgrid_io.parse_textgrid(LONG[:400])


def test_utf16_grid(tmp_path):
  path = tmp_path / "a.TextGrid"
  path.write_bytes(SHORT.encode("utf-16"))
  assert textgrid_io.read_textgrid(path)[1].marks == ("s7", "s8", "s9")


def test_synthesize_grids_end_to_end(tmp_path):
  from oracle_shim import oracle_synthesizer
  from tacotron_amd.checkpoint import load_checkpoint
  ckpt = os.path.join(GOLDEN, "tiny_checkpoint.pt")
  src = tmp_path / "grids"
  (src / "sub").mkdir(parents=True)
  (src / "one.TextGrid").write_text(LONG, encoding="UTF-8")
  (src / "sub" / "two.textgrid").write_text(SHORT, encoding="UTF-8")
  (src / "sub" / "broken.TextGrid").write_text("nothing", encoding="UTF-8")
  (src / "sub" / "other.txt").write_text("x", encoding="UTF-8")
  out = tmp_path / "out"
  ns = cli.build_parser().parse_args(["synthesize-grids", ckpt, str(src), "symbols", "--seed", "5", "--max-decoder-steps", "15",
                                      "--speaker", "speaker1;en", "-out", str(out)])
  assert ns.handler(ns, synthesizer_factory=oracle_synthesizer)
  files = sorted(str(p.relative_to(out)) for p in out.rglob("*") if p.is_file())
  assert files == ["one.npy", "sub/two.npy"]
  synth = oracle_synthesizer(load_checkpoint(ckpt))
  want = synth.infer(("s1", "s12", "s3", "s30"), "speaker1", max_decoder_steps=15, seed=5)
  np.testing.assert_array_equal(np.load(out / "one.npy"), want.mel_outputs_postnet)
  want = synth.infer(("s7", "s8", "s9"), "speaker1", max_decoder_steps=15, seed=5)
  np.testing.assert_array_equal(np.load(out / "sub/two.npy"), want.mel_outputs_postnet)
  # a tier that no grid has: nothing is written, the command still succeeds; an unknown speaker fails
  ns = cli.build_parser().parse_args(["synthesize-grids", ckpt, str(src), "phones", "--seed", "5", "-out", str(tmp_path / "o2")])
  assert ns.handler(ns, synthesizer_factory=oracle_synthesizer)
  assert not (tmp_path / "o2").exists() or not any((tmp_path / "o2").rglob("*.npy"))
  ns = cli.build_parser().parse_args(["synthesize-grids", ckpt, str(src), "symbols", "--speaker", "nobody"])
  assert not ns.handler(ns, synthesizer_factory=oracle_synthesizer)
