"""``synthesize``: every line of a text file -> one mel spectrogram ``.npy`` file;
``synthesize-grids``: every ``.TextGrid`` file of a folder -> one ``.npy`` file.

    python -m tacotron_amd.cli synthesize CHECKPOINT TEXT [options]
    python -m tacotron_amd.cli synthesize-grids CHECKPOINT DIRECTORY TIER [options]

Mirrors the flags, line/paragraph numbering and output naming of the reference's ``synthesize``
command (reference: src/tacotron_cli/inference.py:31-306), so that tools consuming its output
directory (e.g. a vocoder reading ``<line>-<utterance>.npy`` holding fp32 ``[n_mel, T]``) work
unchanged.  Additions: ``--batch-size`` (the reference left that flag commented out,
inference.py:72), ``--precision`` (fp32 as the reference, or the bf16 setting of the decoder of
batches) and multi-GPU sharding when started with one process per GPU
(``python -m torch.distributed.run --nproc-per-node N -m tacotron_amd.cli synthesize ...``):
the lines are partitioned over the ranks, every rank writes its own files, rank 0 reports.

With ``--include-stats`` the reference writes PNG plots next to a ``.log`` file; plotting is not
part of this package: the same ``.log`` is written, and the arrays the plots were drawn from are
saved as ``<stem>-1-alignments.npy`` and ``<stem>-2-prepost.npy``.
"""
from __future__ import annotations

import argparse
import codecs
import collections
import concurrent.futures
import datetime
import logging
import os
import random
import sys
from collections import OrderedDict
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Set, Tuple

import numpy as np

from tacotron_amd import sharding
from tacotron_amd.config import parse_hparams_string

DEFAULT_MAX_DECODER_STEPS = 3000     # reference: src/tacotron_cli/defaults.py
DEFAULT_DEVICE = "cuda:0"
WRITE_BEHIND = 16                     # lines whose files may still be on their way to the disk (batch mode: one batch's worth)
WRITERS = 2                           # threads writing them (numpy releases the interpreter lock while it writes)

Utterances = "OrderedDict[int, List[str]]"
Paragraphs = "OrderedDict[int, OrderedDict[int, List[str]]]"


def split_symbols(line: str, sep: str) -> List[str]:
  """An empty separator means one symbol per character (reference: inference.py:31-34)."""
  return list(line) if sep == "" else line.split(sep)


def parse_paragraphs_from_text(text: str, sep: str) -> "OrderedDict[int, OrderedDict[int, List[str]]]":
  """Blank lines separate paragraphs (numbered from 1); every non-blank line is an utterance keyed
  by its 1-based line number in the file (reference: inference.py:37-56)."""
  paragraphs: "OrderedDict[int, OrderedDict[int, List[str]]]" = OrderedDict()
  current: "OrderedDict[int, List[str]]" = OrderedDict()
  for line_nr, line in enumerate(text.splitlines(), start=1):
    if line == "":
      if current:
        paragraphs[len(paragraphs) + 1] = current
        current = OrderedDict()
      continue
    current[line_nr] = split_symbols(line, sep)
  if current:
    paragraphs[len(paragraphs) + 1] = current
  return paragraphs


@dataclass
class PlannedLine:
  line_nr: int
  utterance_nr: int
  symbols: List[str]
  folder: Path
  stem: str

  @property
  def mel_path(self) -> Path:
    return self.folder / f"{self.stem}.npy"

  @property
  def log_path(self) -> Path:
    return self.folder / f"{self.stem}.log"

  @property
  def alignments_path(self) -> Path:
    return self.folder / f"{self.stem}-1-alignments.npy"

  @property
  def prepost_path(self) -> Path:
    return self.folder / f"{self.stem}-2-prepost.npy"


def plan_outputs(paragraphs, lines_to_infer: Sequence[int], output_directory: Path, paragraph_directories: bool,
                 prepend: str, append: str, overwrite: bool, include_stats: bool,
                 log: Optional[logging.Logger] = None) -> List[PlannedLine]:
  """The lines that will be synthesized and where each goes (reference: inference.py:158-232).

  File stem = ``<prepend><line_nr>-<utterance_nr><append>`` with both numbers zero-filled to the
  width of the largest line number / the utterance count.  The utterance counter follows the
  reference exactly, including its quirk that a line skipped because its output already exists does
  not advance the counter while a line skipped by ``--custom-lines`` does.  Paragraph folders are
  ``<paragraph>-<first line>-<last line>``.  Existence checks happen here, before any synthesis:
  equivalent to the reference's interleaved checks because no line's outputs share a name with
  another line's."""
  log = log or logging.getLogger(__name__)
  wanted = set(lines_to_infer)
  width_par = len(str(max(paragraphs.keys())))
  width_line = len(str(max(nr for par in paragraphs.values() for nr in par.keys())))
  width_utt = len(str(sum(len(par) for par in paragraphs.values())))
  plan: List[PlannedLine] = []
  utterance_nr = 1
  for paragraph_nr, utterances in paragraphs.items():
    folder = output_directory
    if paragraph_directories:
      first, last = min(utterances.keys()), max(utterances.keys())
      folder = output_directory / (f"{str(paragraph_nr).zfill(width_par)}-{str(first).zfill(width_line)}-"
                                   f"{str(last).zfill(width_line)}")
    for line_nr, symbols in utterances.items():
      if line_nr not in wanted:
        log.debug(f"Skipped line {line_nr}.")
        utterance_nr += 1
        continue
      stem = f"{prepend}{str(line_nr).zfill(width_line)}-{str(utterance_nr).zfill(width_utt)}{append}"
      item = PlannedLine(line_nr, utterance_nr, symbols, folder, stem)
      if not overwrite:
        if item.mel_path.exists():
          log.info(f"Line {line_nr}: Skipped inference because line is already synthesized!")
          continue
        if include_stats:
          existing = [p for p in (item.log_path, item.alignments_path, item.prepost_path) if p.exists()]
          if existing:
            log.info(f"Line {line_nr}: {existing[0].name} already exists! Skipped inference.")
            continue
      plan.append(item)
      utterance_nr += 1
  return plan


def _stats_log(res) -> str:
  lines = [
    f"Timepoint: {datetime.datetime.now()}",
    f"Reached max decoder steps: {res.reached_max_decoder_steps}",
    f"Inference duration: {res.inference_duration_s}",
    f"Sampling rate: {res.sampling_rate}",
    f"Spectrogram duration: {res.duration_s:.2f}s",
  ]
  if res.unmappable_symbols:
    lines.append(f"Unknown symbols: {' '.join(sorted(res.unmappable_symbols))}")
  else:
    lines.append("All symbols were known.")
  return "\n".join(lines)


def _write_line(item: PlannedLine, res, include_stats: bool) -> Dict[str, object]:
  item.folder.mkdir(parents=True, exist_ok=True)
  np.save(item.mel_path, res.mel_outputs_postnet)
  if include_stats:
    item.log_path.write_text(_stats_log(res), encoding="UTF-8")
    np.save(item.alignments_path, res.alignments)
    np.save(item.prepost_path, res.mel_outputs)
  return dict(line_nr=item.line_nr, path=str(item.mel_path), frames=int(res.mel_outputs_postnet.shape[-1]),
              duration_s=float(res.duration_s), reached_max=bool(res.reached_max_decoder_steps),
              inference_s=float(res.inference_duration_s),
              unmappable=sorted(res.unmappable_symbols) if res.unmappable_symbols else [])


def synthesize_plan(synth, plan: Sequence[PlannedLine], speaker: Optional[str], seed: int, max_decoder_steps: int,
                    include_stats: bool, batch_size: int, group=None) -> Optional[List[Dict[str, object]]]:
  """Synthesizes the planned lines, sharded over the ranks of the process group if there is one;
  returns the per-line records in plan order on rank 0 (``None`` on other ranks)."""
  costs = [sharding.predicted_cost(len(item.symbols)) for item in plan]

  def work(indices: List[int]) -> Dict[int, Dict[str, object]]:
    # The files of a line are written by background threads while the GPU works on the next line (or batch);
    # at most WRITE_BEHIND lines (a batch's worth in batch mode, so that one batch is written while the next
    # is decoded) wait for the disk, and a failed write surfaces at its own line.
    pending: "collections.deque" = collections.deque()
    out: Dict[int, Dict[str, object]] = {}

    def settle(limit: int) -> None:
      while len(pending) > limit:
        i, future = pending.popleft()
        out[i] = future.result()

    groups = list(sharding.batches(indices, batch_size))
    # The stages below are threads of one interpreter.  A thread that comes back from a library call (this one, after
    # every GPU call) waits for the interpreter lock until the thread holding it is asked to let go, by default after
    # 5 ms - several times per batch, with the GPU idle.  A shorter interval makes those hand-overs prompt.
    switch_interval = sys.getswitchinterval()
    if batch_size > 1:
      sys.setswitchinterval(min(switch_interval, 2e-4))
    try:
      return staged(groups, pending, out, settle)
    finally:
      sys.setswitchinterval(switch_interval)

  def staged(groups, pending, out, settle) -> Dict[int, Dict[str, object]]:
    with concurrent.futures.ThreadPoolExecutor(max_workers=WRITERS, thread_name_prefix="taco-write") as writer, \
         concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="taco-prep") as prep, \
         concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="taco-finish") as finisher:
      # Batch mode is a pipeline of three stages.  This thread keeps the GPU busy: it launches a batch (encoder, decoder,
      # postnet, copies to the host enqueued) and goes straight on to the next one.  A second thread has mapped that
      # batch's symbols to ids meanwhile, and a third waits for a launched batch's copies, cuts it into lines, has the
      # writer threads store them and gives the batch's staging buffers back.  The library call releases the
      # interpreter lock for as long as the decoder runs, so the other stages run beside it.
      pipelined = batch_size > 1 and hasattr(synth, "launch_batch") and hasattr(synth, "prepare_batch")
      ahead = None
      if pipelined and groups:
        ahead = prep.submit(synth.prepare_batch, [plan[i].symbols for i in groups[0]], speaker)
      finishing: "collections.deque" = collections.deque()

      def finish(launched) -> Dict[int, Dict[str, object]]:
        results = synth.finish_batch(launched)
        written = [writer.submit(_write_line, plan[i], res, include_stats) for i, res in zip(launched.finished_tags, results)]
        records = {i: f.result() for i, f in zip(launched.finished_tags, written)}
        launched.release()                   # the lines' arrays were views into the batch's staging buffers
        return records

      # The decoder of batches keeps sixteen lines in flight; a batch that is decoded to its last line ends with most of
      # those slots empty.  So every launch but the last stops as soon as a slot has no successor, and the lines still
      # running head the next launch, where they go on from their next frame (same frames as if never interrupted).
      resumable = pipelined and max_decoder_steps > 0 and getattr(getattr(synth, "model", None), "supports_resume", False)
      carry = None

      for g, group_idx in enumerate(groups):
        if pipelined:
          prepared = ahead.result()
          ahead = prep.submit(synth.prepare_batch, [plan[i].symbols for i in groups[g + 1]], speaker) if g + 1 < len(groups) else None
          launched = synth.launch_batch([plan[i].symbols for i in group_idx], speaker=speaker, max_decoder_steps=max_decoder_steps,
                                        seed=seed, include_stats=include_stats, prepared=prepared, tags=group_idx, carry=carry,
                                        leave_unfinished=resumable and g + 1 < len(groups))
          carry = launched.unfinished
          finishing.append(finisher.submit(finish, launched))
          while len(finishing) > 2:          # (a batch's staging set is free again when its lines have been written)
            out.update(finishing.popleft().result())
          continue
        if batch_size == 1:
          results = [synth.infer(symbols=plan[group_idx[0]].symbols, speaker=speaker, include_stats=include_stats,
                                 max_decoder_steps=max_decoder_steps, seed=seed)]
        else:
          results = synth.infer_batch([plan[i].symbols for i in group_idx], speaker=speaker, max_decoder_steps=max_decoder_steps,
                                      seed=seed, include_stats=include_stats)
        for i, res in zip(group_idx, results):
          pending.append((i, writer.submit(_write_line, plan[i], res, include_stats)))
        settle(max(WRITE_BEHIND, batch_size))
      while finishing:
        out.update(finishing.popleft().result())
      settle(0)
    return out

  return sharding.run_sharded(len(plan), costs, work, group=group)


# ---- argument parsing (types follow the reference: src/tacotron_cli/argparse_helper.py) ------------
def _existing_file(value: str) -> Path:
  path = Path(value)
  if not path.is_file():
    raise argparse.ArgumentTypeError("File was not found!")
  return path


def _codec(value: str) -> str:
  try:
    codecs.lookup(value)
  except LookupError as error:
    raise argparse.ArgumentTypeError("Codec was not found!") from error
  return value


def _non_empty(value: str) -> str:
  if value == "":
    raise argparse.ArgumentTypeError("Value must not be empty!")
  return value


def _integer(value: str, minimum: int, what: str) -> int:
  if not value.isdigit():
    raise argparse.ArgumentTypeError("Value needs to be an integer!")
  number = int(value)
  if number < minimum:
    raise argparse.ArgumentTypeError(f"Value needs to be {what}!")
  return number


def _non_negative_integer(value: str) -> int:
  return _integer(value, 0, "greater than or equal to zero")


def _positive_integer(value: str) -> int:
  return _integer(value, 1, "greater than zero")


def init_synthesis_parser(parser: argparse.ArgumentParser) -> Callable[[argparse.Namespace], bool]:
  parser.description = "Synthesize each line of an text file into a mel-spectrogram."
  parser.add_argument("checkpoint", metavar="CHECKPOINT", type=_existing_file,
                      help="path to checkpoint that should be used for synthesis")
  parser.add_argument("text", metavar="TEXT", type=_existing_file,
                      help="path to text file containing lines that should be synthesized")
  parser.add_argument("--sep", type=str, default="", help="separator between each symbol")
  parser.add_argument("--encoding", type=_codec, default="UTF-8", help="encoding of TEXT")
  parser.add_argument("--custom-speaker", type=_non_empty, default=None,
                      help="use that speaker for syntheses; defaults to the first speaker if left unset")
  parser.add_argument("--custom-lines", type=_non_negative_integer, nargs="*", default=[],
                      help="synthesize only lines with these numbers")
  parser.add_argument("--max-decoder-steps", type=_positive_integer, metavar="MAX-DECODER-STEPS",
                      default=DEFAULT_MAX_DECODER_STEPS, help="maximum step count before synthesis is stopped")
  parser.add_argument("--batch-size", type=_positive_integer, default=1,
                      help="lines synthesized together in one padded batch (1 = line by line, as the reference)")
  parser.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                      help="arithmetic of the decoder of batches: fp32 (as the reference) or bf16 gate products (faster, "
                           "frames within 2e-3 of fp32 under teacher forcing; needs --batch-size > 1 to matter)")
  parser.add_argument("--custom-seed", type=_non_negative_integer, default=None,
                      help="custom seed used for synthesis; if left unset a random seed will be chosen")
  parser.add_argument("-p", "--paragraph-directories", action="store_true",
                      help="create a new directory for each paragraph")
  parser.add_argument("--include-stats", action="store_true",
                      help="also write a log, the alignments and the pre-postnet spectrogram per line")
  parser.add_argument("--device", type=str, metavar="DEVICE", default=DEFAULT_DEVICE,
                      help="use this device, e.g., \"cuda:0\"; with one process per GPU the local rank selects it")
  parser.add_argument("--custom-hparams", type=_non_empty, metavar="CUSTOM-HYPERPARAMETERS", default=None,
                      help="custom hyperparameters comma separated")
  parser.add_argument("--prepend", type=str, default="", help="prepend this text to all output files")
  parser.add_argument("--append", type=str, default="", help="append this text to all output files")
  parser.add_argument("-out", "--output-directory", type=Path, default=None,
                      help="custom output directory if parent folder of TEXT should not be used")
  parser.add_argument("-o", "--overwrite", action="store_true", help="overwrite already synthesized lines")
  return synthesize_ns


def _init_ranks(device: str):
  """One process per GPU when started by torch.distributed.run: a host-only (gloo) group carries
  the gather of the per-line records; the GPU is picked by LOCAL_RANK."""
  n = int(os.environ.get("WORLD_SIZE", "1"))
  if n <= 1:
    return device, None
  import torch.distributed as dist
  if not dist.is_initialized():
    dist.init_process_group("gloo")
  return f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}", dist.group.WORLD


def synthesize_ns(ns: argparse.Namespace, synthesizer_factory: Optional[Callable] = None,
                  checkpoint_loader: Optional[Callable] = None) -> bool:
  """Runs the command; returns success like the reference's handler (inference.py:92-306).
  ``synthesizer_factory(checkpoint, custom_hparams, device)`` and ``checkpoint_loader(path)`` default
  to this package's HIP-backed ``Synthesizer`` and ``load_checkpoint``."""
  log = logging.getLogger(__name__)
  if checkpoint_loader is None:
    from tacotron_amd.checkpoint import load_checkpoint as checkpoint_loader
  try:
    checkpoint = checkpoint_loader(ns.checkpoint)
  except Exception as ex:  # pylint: disable=broad-except
    log.debug(ex)
    log.error("Checkpoint couldn't be loaded!")
    return False

  speaker_mapping = checkpoint.get("speaker_mapping")
  if ns.custom_speaker is not None and (speaker_mapping is None or ns.custom_speaker not in speaker_mapping):
    log.error("Custom speaker was not found!")
    return False
  try:
    text = Path(ns.text).read_text(ns.encoding)
  except Exception:  # pylint: disable=broad-except
    log.error("Text couldn't be read!")
    return False

  output_directory = ns.output_directory
  if output_directory is None:
    output_directory = Path(ns.text).parent / Path(ns.text).stem
  if output_directory.is_file():
    log.error("Output directory is a file!")
    return False

  paragraphs = parse_paragraphs_from_text(text, ns.sep)
  if len(paragraphs) == 0:
    log.error("Text contains no lines!")
    return False
  all_lines = [nr for par in paragraphs.values() for nr in par.keys()]
  lines_to_infer: List[int] = all_lines
  custom_lines = list(OrderedDict.fromkeys(ns.custom_lines))
  if custom_lines:
    for nr in custom_lines:
      if nr not in all_lines:
        log.error(f"Line {nr} is not inferable!")
        return False
    lines_to_infer = custom_lines

  device, group = _init_ranks(ns.device)
  rank, n_ranks = sharding.world()
  if ns.custom_seed is not None:
    seed = ns.custom_seed
  else:
    seed = random.randint(1, 9999)
    if n_ranks > 1:                          # every rank must use the same seed
      import torch.distributed as dist
      box = [seed]
      dist.broadcast_object_list(box, src=0, group=group)
      seed = box[0]
    log.info(f"Using random seed: {seed}.")

  if ns.custom_speaker is not None:
    speaker = ns.custom_speaker
  else:
    speaker = next(iter(speaker_mapping.keys())) if speaker_mapping else None
    log.debug(f"Speaker: {speaker}")

  # The plan depends on which output files exist.  With several ranks it is made once, on rank 0, before any
  # rank can have written a file, and handed to the others: ranks that planned for themselves could see each
  # other's first outputs as "already synthesized" and disagree about the shards.
  if n_ranks > 1:
    import torch.distributed as dist
    box = [None]
    if rank == 0:
      # whatever rank 0 runs into here (an unreadable output folder, say) travels to the others in place of the plan:
      # they would otherwise wait in the broadcast until the process group's timeout
      try:
        box = [("plan", plan_outputs(paragraphs, lines_to_infer, output_directory, ns.paragraph_directories, ns.prepend,
                                     ns.append, ns.overwrite, ns.include_stats, log))]
      except Exception as ex:  # pylint: disable=broad-except
        box = [("error", f"{type(ex).__name__}: {ex}")]
    dist.broadcast_object_list(box, src=0, group=group)
    kind, plan = box[0]
    if kind == "error":
      log.error(f"Planning the output files failed on rank 0: {plan}")
      return False
  else:
    plan = plan_outputs(paragraphs, lines_to_infer, output_directory, ns.paragraph_directories, ns.prepend, ns.append,
                        ns.overwrite, ns.include_stats, log)

  if synthesizer_factory is None:
    from tacotron_amd.synthesizer import Synthesizer as synthesizer_factory
  synth = synthesizer_factory(checkpoint, parse_hparams_string(ns.custom_hparams), device)
  if getattr(ns, "precision", "fp32") != "fp32":
    synth.model.set_precision(ns.precision)
  log.info("Inferring...")
  records = synthesize_plan(synth, plan, speaker, seed, ns.max_decoder_steps, ns.include_stats, ns.batch_size, group)
  if records is None:                        # not rank 0
    return True
  unknown: Set[str] = set()
  total_s = 0.0
  for rec in records:
    log.info(f"Line {rec['line_nr']}: spectrogram duration: {rec['duration_s']:.2f}s")
    total_s += rec["duration_s"]
    unknown.update(rec["unmappable"])
  if unknown:
    log.warning(f"Unknown symbols: {' '.join(sorted(unknown))} (#{len(unknown)})")
  log.info(f"Done. Total spectrogram duration: {total_s:.2f}s")
  log.info(f"Written output to: '{output_directory.absolute()}'")
  return True


# ---- synthesize-grids: one spectrogram per .TextGrid file (reference: src/tacotron_cli/textgrid_inference.py) ----
def _existing_directory(value: str) -> Path:
  path = Path(value)
  if not path.is_dir():
    raise argparse.ArgumentTypeError("Directory was not found!")
  return path


def init_grid_synthesis_parser(parser: argparse.ArgumentParser) -> Callable[[argparse.Namespace], bool]:
  parser.description = "Synthesize each .TextGrid file into a mel-spectrogram."
  parser.add_argument("checkpoint", metavar="CHECKPOINT", type=_existing_file,
                      help="path to checkpoint that should be used for synthesis")
  parser.add_argument("directory", metavar="DIRECTORY", type=_existing_directory,
                      help="path to folder containing .TextGrid files that should be synthesized")
  parser.add_argument("tier", metavar="TIER", type=_non_empty, help="tier containing the symbols in separate intervals")
  parser.add_argument("--speaker", type=_non_empty, default=None,
                      help="use that speaker for syntheses; defaults to the first speaker if left unset")
  parser.add_argument("--max-decoder-steps", type=_positive_integer, metavar="MAX-DECODER-STEPS",
                      default=DEFAULT_MAX_DECODER_STEPS, help="maximum step count before synthesis is stopped")
  parser.add_argument("--batch-size", type=_positive_integer, default=1,
                      help="grids synthesized together in one padded batch (1 = one by one, as the reference)")
  parser.add_argument("--precision", choices=("fp32", "bf16"), default="fp32",
                      help="arithmetic of the decoder of batches: fp32 (as the reference) or bf16 gate products (faster, "
                           "frames within 2e-3 of fp32 under teacher forcing; needs --batch-size > 1 to matter)")
  parser.add_argument("--seed", type=_non_negative_integer, default=None,
                      help="custom seed used for synthesis; if left unset a random seed will be chosen")
  parser.add_argument("--device", type=str, metavar="DEVICE", default=DEFAULT_DEVICE,
                      help="use this device, e.g., \"cuda:0\"; with one process per GPU the local rank selects it")
  parser.add_argument("--custom-hparams", type=_non_empty, metavar="CUSTOM-HYPERPARAMETERS", default=None,
                      help="custom hyperparameters comma separated")
  parser.add_argument("--encoding", type=_codec, default="UTF-8", help="encoding of .TextGrid files")
  parser.add_argument("-out", "--output-directory", type=Path, default=None, help="custom output directory")
  return synthesize_grids_ns


def synthesize_grids_ns(ns: argparse.Namespace, synthesizer_factory: Optional[Callable] = None,
                        checkpoint_loader: Optional[Callable] = None) -> bool:
  """Every ``*.TextGrid`` below DIRECTORY (any depth, suffix compared case-insensitively) becomes
  ``<same relative folder>/<stem>.npy`` below the output directory; the symbols are the interval
  marks of the first tier called TIER.  Unreadable grids and grids without that tier are reported
  and skipped (reference: textgrid_inference.py:59-167)."""
  from tacotron_amd import textgrid_io
  log = logging.getLogger(__name__)
  if checkpoint_loader is None:
    from tacotron_amd.checkpoint import load_checkpoint as checkpoint_loader
  try:
    checkpoint = checkpoint_loader(ns.checkpoint)
  except Exception as ex:  # pylint: disable=broad-except
    log.debug(ex)
    log.error("Checkpoint couldn't be loaded!")
    return False

  speaker_mapping = checkpoint.get("speaker_mapping") or {}
  if ns.speaker is not None:
    speaker = ns.speaker
    if speaker not in speaker_mapping:
      speaker = ns.speaker.split(";")[0]        # "name;accent"-style ids fall back to their first part
      if speaker not in speaker_mapping:
        log.error(f"Speaker \"{ns.speaker}\" doesn't exist!")
        return False
  else:
    speaker = next(iter(speaker_mapping.keys())) if speaker_mapping else None
  log.info(f"Speaker: \"{speaker}\"")

  output_directory = ns.output_directory if ns.output_directory is not None else ns.directory
  if output_directory.is_file():
    log.error("Output directory is a file!")
    return False

  device, group = _init_ranks(ns.device)
  rank, n_ranks = sharding.world()
  seed = ns.seed
  if seed is None:
    seed = random.randint(1, 9999)
    if n_ranks > 1:
      import torch.distributed as dist
      box = [seed]
      dist.broadcast_object_list(box, src=0, group=group)
      seed = box[0]
  log.info(f"Seed: {seed}")

  grid_paths = sorted(Path(root) / name for root, _, files in os.walk(ns.directory) for name in files
                      if Path(name).suffix.lower() == ".textgrid")
  plan: List[PlannedLine] = []
  for nr, grid_path in enumerate(grid_paths, start=1):
    try:
      tiers = textgrid_io.read_textgrid(grid_path, ns.encoding)
    except Exception as ex:  # pylint: disable=broad-except
      log.error(f"File \"{grid_path.absolute()}\" couldn't be read! Skipped.")
      log.debug(ex)
      continue
    tier = textgrid_io.first_tier(tiers, ns.tier)
    if tier is None or tier.tier_class != "IntervalTier":
      log.error(f"File \"{grid_path.absolute()}\" has no tier \"{ns.tier}\"! Skipped.")
      continue
    folder = output_directory / grid_path.relative_to(ns.directory).parent
    plan.append(PlannedLine(nr, nr, list(tier.marks), folder, grid_path.stem))

  if synthesizer_factory is None:
    from tacotron_amd.synthesizer import Synthesizer as synthesizer_factory
  synth = synthesizer_factory(checkpoint, parse_hparams_string(ns.custom_hparams), device)
  if getattr(ns, "precision", "fp32") != "fp32":
    synth.model.set_precision(ns.precision)
  log.info("Inferring...")
  records = synthesize_plan(synth, plan, speaker, seed, ns.max_decoder_steps, False, ns.batch_size, group)
  if records is None:
    return True
  unknown: Set[str] = set()
  for rec in records:
    log.info(f"Saved output to: \"{rec['path']}\" (spectrogram duration: {rec['duration_s']:.2f}s)")
    if rec["reached_max"]:
      log.warning("Reached max decoder steps!")
    unknown.update(rec["unmappable"])
  if unknown:
    log.warning(f"Unknown symbols: {' '.join(sorted(unknown))} (#{len(unknown)})")
  log.info(f"Written output to: \"{output_directory.absolute()}\"")
  return True


def build_parser() -> argparse.ArgumentParser:
  parser = argparse.ArgumentParser(prog="tacotron-amd", description="Tacotron 2 mel synthesis on MI355X")
  sub = parser.add_subparsers(dest="command", required=True)
  handler = init_synthesis_parser(sub.add_parser("synthesize"))
  sub.choices["synthesize"].set_defaults(handler=handler)
  handler = init_grid_synthesis_parser(sub.add_parser("synthesize-grids"))
  sub.choices["synthesize-grids"].set_defaults(handler=handler)
  parser.add_argument("-v", "--verbose", action="store_true")
  return parser


def main(argv: Optional[Sequence[str]] = None) -> int:
  ns = build_parser().parse_args(argv)
  logging.basicConfig(level=logging.DEBUG if ns.verbose else logging.INFO, format="%(levelname)s %(message)s")
  ok = ns.handler(ns)
  try:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
      dist.destroy_process_group()
  except Exception:  # pylint: disable=broad-except
    pass
  return 0 if ok else 1


if __name__ == "__main__":
  sys.exit(main())
