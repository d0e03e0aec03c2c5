#!/usr/bin/env python3
"""The last N kernel dispatches of one `rocprofv3 --kernel-trace` output directory, one per line: duration, grid, name.
Usage: kernel_list.py <rocprofv3 output dir> [N]"""
import sqlite3, sys
from pathlib import Path

found = sorted(Path(sys.argv[1]).rglob("*_results.db"))
if not found:
  sys.exit(f"no database under {sys.argv[1]}")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
con = sqlite3.connect(found[0])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
grid = [c for c in ("grid_x", "grid_y", "grid_z", "workgroup_x") if c in cols]
order = "start" if "start" in cols else "rowid"
rows = con.execute(f"select duration, name{''.join(', ' + c for c in grid)} from kernels order by {order} desc limit {n}").fetchall()
for r in reversed(rows):
  print(f"{r[0] / 1000.0:9.1f} us  grid {tuple(r[2:])}  {r[1][:70]}")
