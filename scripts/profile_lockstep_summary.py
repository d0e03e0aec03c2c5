#!/usr/bin/env python3
"""Summaries of one scripts/profile_lockstep.sh pass, written under profiles/ as <tag>_lockstep_*.

  <tag>_lockstep_batch64.txt        the untraced throughput line
  <tag>_lockstep_kernel_stats.csv   per-kernel calls / total / average from the kernel trace
  <tag>_lockstep_pmc.json           counters of taco_decoder_batch_kernel per dispatch (one dispatch = the batch of 64:
                                    2400 lockstep frames) and what follows from them: matrix-core busy share, issue
                                    fractions, HBM bytes per lockstep frame
"""
import csv
import json
import shutil
import sqlite3
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
src = root / "gpurun_out" / f"prof_{tag}_lockstep"
dst = root / "profiles"
FRAMES = 64 * 600 // 16     # lockstep frames of the one dispatch that decodes the batch: 64 utterances x 600 frames, 16 in flight
KERNEL = "taco_decoder_batch_kernel"
SIMDS = 256 * 4


def db(path: Path):
  found = sorted(path.rglob("*_results.db"))
  return sqlite3.connect(found[0]) if found else None


shutil.copy(src / "batch64.txt", dst / f"{tag}_lockstep_batch64.txt")
con = db(src / "kt")
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                   "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
kernel_ns = None
with open(dst / f"{tag}_lockstep_kernel_stats.csv", "w", newline="") as f:
  w = csv.writer(f)
  w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
  for name, calls, tot, avg, lo, hi in rows:
    w.writerow([name[:100], calls, tot, round(avg, 1), round(100.0 * tot / total, 2), lo, hi])
    if KERNEL in name:
      kernel_ns = avg

c = {}
for sub in ("sqa", "sqm", "sqb", "fetch", "write"):
  con = db(src / sub)
  if con is None:
    continue
  q = ("select counter_name, avg(v), count(*) from (select counter_name, dispatch_id, sum(value) as v from counters_collection "
       f"where kernel_name like '%{KERNEL}%' group by counter_name, dispatch_id) group by counter_name")
  for cname, avg, n in con.execute(q):
    c[cname] = avg
    c["_dispatches_" + sub] = n

d = {}
wc = c.get("SQ_WAVE_CYCLES")
if wc:
  for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_ANY"):
    if k in c:
      d[k + "_per_WAVE_CYCLE"] = round(c[k] / wc, 4)
for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_F32", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_BF16",
          "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
          "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
  if k in c:
    d[k + "_per_lockstep_frame"] = round(c[k] / FRAMES, 1)
if kernel_ns and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
  # the counter sums, over all SIMDs, the cycles in which a SIMD's matrix core is busy; the kernel holds
  # all 1024 SIMDs for its whole duration at the shader clock (2.4 GHz)
  cycles_available = kernel_ns * 2.4 * SIMDS
  d["mfma_busy_share_of_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / cycles_available, 4)
  n_mfma = c.get("SQ_INSTS_VALU_MFMA_F32", 0) + c.get("SQ_INSTS_VALU_MFMA_BF16", 0) or c.get("SQ_INSTS_MFMA", 1)
  d["mfma_busy_cycles_per_mfma_instruction"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(n_mfma, 1), 2)
if "SQ_LDS_IDX_ACTIVE" in c:
  d["lds_bank_conflict_cycles_per_active_cycle"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
  d["hbm_bytes_per_lockstep_frame"] = int(round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / FRAMES))
  d["traffic_correction"] = "FETCH_SIZE x2 (gfx950 reports half), WRITE_SIZE as reported, KiB units, separate passes"
(dst / f"{tag}_lockstep_pmc.json").write_text(json.dumps({
  "kernel": KERNEL, "precision": __import__("os").environ.get("TACO_PRECISION", "fp32"),
  "workload": "64 lines of 40..160 symbols, 600 frames each: one dispatch, 16 utterances in flight, slots refilled",
  "lockstep_frames_per_dispatch": FRAMES, "kernel_avg_ns": kernel_ns,
  "per_dispatch": {k: round(v, 1) for k, v in sorted(c.items())}, "derived": d,
  "command": "rocprofv3 --pmc <one block-limited set per pass> -- python3 scripts/batch_once.py 64 600 1 (scripts/profile_lockstep.sh)",
}, indent=1) + "\n")
print(json.dumps(d, indent=1))
