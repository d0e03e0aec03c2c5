#!/usr/bin/env python3
"""Summary of one scripts/profile_conv.sh pass: counters of the longest dispatches of taco_conv_kernel<5, false>
(the postnet's 512 -> 512 layers), printed and written to profiles/<tag>_conv_pmc.json."""
import json
import sqlite3
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
src = root / "gpurun_out" / f"prof_{tag}_conv"
KERNEL = "taco_conv_kernel<5, false>"
SIMDS = 256 * 4


def db(path: Path):
  found = sorted(path.rglob("*_results.db"))
  return sqlite3.connect(found[0]) if found else None


con = db(src / "kt")
durs = sorted(r[0] for r in con.execute(f"select duration from kernels where name like '%{KERNEL}%'"))
# three of the five layers are 512 -> 512: the upper 60 % of the durations
wide = durs[int(len(durs) * 0.45):]
kernel_ns = sum(wide) / len(wide)
c = {}
for sub in ("sqa", "sqm", "sqb"):
  con = db(src / sub)
  if con is None:
    continue
  rows = con.execute("select counter_name, dispatch_id, sum(value) from counters_collection "
                     f"where kernel_name like '%{KERNEL}%' group by counter_name, dispatch_id").fetchall()
  per = {}
  for name, disp, v in rows:
    per.setdefault(name, []).append(v)
  for name, vs in per.items():
    vs.sort()
    top = vs[int(len(vs) * 0.45):]
    c[name] = sum(top) / len(top)
d = {"kernel_avg_ns_wide_layers": round(kernel_ns, 1)}
wc = c.get("SQ_WAVE_CYCLES")
if wc:
  for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY"):
    if k in c:
      d[k + "_per_WAVE_CYCLE"] = round(c[k] / wc, 4)
if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
  d["mfma_busy_share_of_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (kernel_ns * 2.4 * SIMDS), 4)
  d["mfma_busy_cycles_per_mfma_instruction"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c.get("SQ_INSTS_MFMA", 1), 1), 2)
if "SQ_LDS_IDX_ACTIVE" in c:
  d["lds_bank_conflict_cycles_per_active_cycle"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
if "SQ_BUSY_CU_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
  d["busy_cu_cycles_per_gui_active"] = round(c["SQ_BUSY_CU_CYCLES"] / c["GRBM_GUI_ACTIVE"], 2)
out = {"kernel": KERNEL, "timing": (src / "timing.txt").read_text().strip(), "per_dispatch_wide_layers": {k: round(v, 1) for k, v in sorted(c.items())},
       "derived": d, "command": "scripts/profile_conv.sh"}
(root / "profiles" / f"{tag}_conv_pmc.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
