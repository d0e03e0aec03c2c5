"""The ``synthesize`` command's host logic: paragraph parsing, output naming, skip rules, and an
end-to-end run with the CPU oracle standing in for the GPU model (tests/oracle_shim.py)."""
import os
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, scenario_checkpoint
from tacotron_amd import cli, scenarios

TINY_CKPT = os.path.join(GOLDEN, "tiny_checkpoint.pt")


def test_parse_paragraphs_line_numbers_and_separators():
  text = "a|b\nc\n\n\nd|e|f\n"
  pars = cli.parse_paragraphs_from_text(text, "|")
  assert list(pars.keys()) == [1, 2]
  assert pars[1] == {1: ["a", "b"], 2: ["c"]}
  assert pars[2] == {5: ["d", "e", "f"]}                      # line numbers count blank lines too
  assert cli.parse_paragraphs_from_text("ab c", "")[1][1] == ["a", "b", " ", "c"]   # "" = per character
  assert cli.parse_paragraphs_from_text("\n\n", "|") == {}


def _plan(pars, lines=None, out=Path("/nonexistent/out"), **kw):
  lines = lines if lines is not None else [nr for p in pars.values() for nr in p]
  args = dict(paragraph_directories=False, prepend="", append="", overwrite=False, include_stats=False)
  args.update(kw)
  return cli.plan_outputs(pars, lines, out, **args)


def test_output_names_are_zero_filled_like_the_reference():
  text = "\n".join(["x"] * 9 + [""] + ["y"] * 3)             # lines 1-9, blank 10, lines 11-13
  pars = cli.parse_paragraphs_from_text(text, "")
  plan = _plan(pars)
  assert [p.stem for p in plan][:2] == ["01-01", "02-02"]
  assert plan[-1].stem == "13-12" and plan[-1].line_nr == 13 and plan[-1].utterance_nr == 12
  assert plan[0].mel_path == Path("/nonexistent/out/01-01.npy")
  plan = _plan(pars, paragraph_directories=True, prepend="pre_", append="_post")
  assert plan[0].mel_path == Path("/nonexistent/out/1-01-09/pre_01-01_post.npy")
  assert plan[-1].mel_path == Path("/nonexistent/out/2-11-13/pre_13-12_post.npy")


def test_custom_lines_keep_global_utterance_numbers():
  pars = cli.parse_paragraphs_from_text("a\nb\n\nc\nd", "")
  plan = _plan(pars, lines=[5, 2])
  assert [(p.line_nr, p.utterance_nr, p.stem) for p in plan] == [(2, 2, "2-2"), (5, 4, "5-4")]


def test_existing_outputs_are_skipped_without_advancing_the_counter(tmp_path):
  pars = cli.parse_paragraphs_from_text("a\nb\nc", "")
  (tmp_path / "1-1.npy").write_bytes(b"")
  plan = _plan(pars, out=tmp_path)
  # the reference's `continue` on an existing file comes before its counter increment
  assert [(p.line_nr, p.stem) for p in plan] == [(2, "2-1"), (3, "3-2")]
  plan = _plan(pars, out=tmp_path, overwrite=True)
  assert [p.stem for p in plan] == ["1-1", "2-2", "3-3"]
  (tmp_path / "2-1.log").write_text("x")
  assert [p.line_nr for p in _plan(pars, out=tmp_path, include_stats=True)] == [3]


def _run(argv, factory):
  ns = cli.build_parser().parse_args(argv)
  return ns.handler(ns, synthesizer_factory=factory)


def test_synthesize_tiny_checkpoint_end_to_end(tmp_path):
  from oracle_shim import oracle_synthesizer
  from tacotron_amd.checkpoint import load_checkpoint
  lines = ["s1|s2|s3|s4|s5|s6", "s7|s8|s9", "", "s10|s11|s12|s13|zz|s14"]
  text = tmp_path / "in.txt"
  text.write_text("\n".join(lines), encoding="UTF-8")
  ok = _run(["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-seed", "5", "--max-decoder-steps", "20",
             "--include-stats", "-p"], oracle_synthesizer)
  assert ok
  out = tmp_path / "in"
  files = sorted(str(p.relative_to(out)) for p in out.rglob("*.npy"))
  assert files == ["1-1-2/1-1-1-alignments.npy", "1-1-2/1-1-2-prepost.npy", "1-1-2/1-1.npy",
                   "1-1-2/2-2-1-alignments.npy", "1-1-2/2-2-2-prepost.npy", "1-1-2/2-2.npy",
                   "2-4-4/4-3-1-alignments.npy", "2-4-4/4-3-2-prepost.npy", "2-4-4/4-3.npy"]
  synth = oracle_synthesizer(load_checkpoint(TINY_CKPT))
  for rel, symbols in (("1-1-2/1-1", lines[0]), ("1-1-2/2-2", lines[1]), ("2-4-4/4-3", lines[3])):
    want = synth.infer(symbols.split("|"), "speaker1", max_decoder_steps=20, seed=5, include_stats=True)
    got = np.load(out / (rel + ".npy"))
    assert got.dtype == np.float32 and got.shape[0] == synth.hparams.n_mel_channels
    np.testing.assert_array_equal(got, want.mel_outputs_postnet)
    np.testing.assert_array_equal(np.load(out / (rel + "-1-alignments.npy")), want.alignments)
    log = (out / (rel + ".log")).read_text()
    assert "Reached max decoder steps:" in log and "Sampling rate: 22050" in log
  assert "Unknown symbols: zz" in (out / "2-4-4/4-3.log").read_text()
  assert np.load(out / "2-4-4/4-3-1-alignments.npy").shape[1] == 5       # the unknown position was dropped
  # second run without -o: everything exists, nothing is rewritten
  stamp = (out / "1-1-2/1-1.npy").stat().st_mtime_ns
  assert _run(["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-seed", "5", "-p"], oracle_synthesizer)
  assert (out / "1-1-2/1-1.npy").stat().st_mtime_ns == stamp


def test_synthesize_batched_equals_line_by_line(tmp_path):
  from oracle_shim import oracle_synthesizer
  lines = ["s1|s2|s3|s4|s5|s6|s7", "s7|s8|s9", "s3|s3|s4|s20|s21", "s30|s29"]
  text = tmp_path / "in.txt"
  text.write_text("\n".join(lines), encoding="UTF-8")
  common = ["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-seed", "5", "--max-decoder-steps", "16"]
  assert _run(common + ["-out", str(tmp_path / "one")], oracle_synthesizer)
  assert _run(common + ["-out", str(tmp_path / "batched"), "--batch-size", "3"], oracle_synthesizer)
  names = sorted(p.name for p in (tmp_path / "one").iterdir())
  assert names == ["1-1.npy", "2-2.npy", "3-3.npy", "4-4.npy"]
  for n in names:
    np.testing.assert_array_equal(np.load(tmp_path / "one" / n), np.load(tmp_path / "batched" / n))


def test_lines_carried_from_launch_to_launch_end_up_in_the_same_files(tmp_path):
  """The batch mode's carry-over (a launch may leave lines unfinished, they head the next launch and finish there or
  later) on the host side: with a model that leaves lines at every launch but the last, the command writes the files it
  writes line by line - every line once, under its own name, with its own frames, --include-stats files included."""
  from oracle_shim import oracle_synthesizer, oracle_synthesizer_resumable
  r = np.random.RandomState(3)
  lines = ["|".join(f"s{r.randint(1, 31)}" for _ in range(r.randint(2, 12))) for _ in range(14)]
  lines.insert(5, "")
  text = tmp_path / "in.txt"
  text.write_text("\n".join(lines), encoding="UTF-8")
  common = ["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-seed", "5", "--max-decoder-steps", "26", "--include-stats"]
  assert _run(common + ["-out", str(tmp_path / "one")], oracle_synthesizer)
  made = []

  def factory(*args, **kwargs):
    made.append(oracle_synthesizer_resumable(*args, **kwargs))
    return made[-1]

  assert _run(common + ["-out", str(tmp_path / "carried"), "--batch-size", "3"], factory)
  assert made[0].model.left_total >= 5 and not made[0].model._parked        # lines were left, and all came back
  names = sorted(p.name for p in (tmp_path / "one").iterdir())
  assert len([n for n in names if n.endswith(".npy")]) == 3 * 14
  assert names == sorted(p.name for p in (tmp_path / "carried").iterdir())
  for n in names:
    if n.endswith(".npy"):
      np.testing.assert_array_equal(np.load(tmp_path / "one" / n), np.load(tmp_path / "carried" / n), err_msg=n)


def test_synthesize_reproduces_the_reference_record(tmp_path, gate_table):
  """North-Wind line through the command = the reference's Synthesizer.infer output for it."""
  from oracle_shim import oracle_synthesizer
  z, meta = load_golden("ljs_synth")
  ck = scenario_checkpoint("ljs", meta["gate"])
  text = tmp_path / "north_wind.txt"
  text.write_text(scenarios.NORTH_WIND_LINE + "\n", encoding="UTF-8")
  ns = cli.build_parser().parse_args(["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-seed",
                                      str(meta["seed"]), "--max-decoder-steps", str(meta["max_steps"])])
  assert ns.handler(ns, synthesizer_factory=oracle_synthesizer, checkpoint_loader=lambda path: ck)
  got = np.load(tmp_path / "north_wind" / "1-1.npy")
  assert got.shape == z["mel_postnet"].shape                  # identical stop frame
  assert np.abs(got - z["mel_postnet"]).max() <= 1e-3         # fp32 free-running tolerance


def test_command_errors_are_reported_not_raised(tmp_path, caplog):
  from oracle_shim import oracle_synthesizer
  text = tmp_path / "in.txt"
  text.write_text("s1|s2\n", encoding="UTF-8")
  assert not _run(["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-speaker", "nobody"], oracle_synthesizer)
  assert not _run(["synthesize", TINY_CKPT, str(text), "--sep", "|", "--custom-lines", "7"], oracle_synthesizer)
  assert not _run(["synthesize", str(text), str(text)], oracle_synthesizer)        # not a checkpoint
  with pytest.raises(SystemExit):
    cli.build_parser().parse_args(["synthesize", TINY_CKPT, str(tmp_path / "missing.txt")])
