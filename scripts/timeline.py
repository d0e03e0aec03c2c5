#!/usr/bin/env python3
"""What the GPU does between two launches of the lockstep decoder: from a `rocprofv3 --kernel-trace --memory-copy-trace`
output directory, the kernels and copies from the end of the n-th decoder launch to the start of the next one, with the
gaps in which the device has nothing to do.  Usage: timeline.py <rocprofv3 output dir> [n=20]"""
import sqlite3, sys
from pathlib import Path

found = sorted(Path(sys.argv[1]).rglob("*_results.db"))
if not found:
  sys.exit(f"no database under {sys.argv[1]}")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
con = sqlite3.connect(found[0])
rows = [(s, e, name) for name, s, e in con.execute("select name, start, end from kernels")]
try:
  rows += [(s, e, f"copy {name} {size} B") for name, s, e, size in con.execute("select name, start, end, size from memory_copies")]
except sqlite3.Error:
  pass
rows.sort()
dec = [i for i, r in enumerate(rows) if "taco_decoder_batch_kernel" in r[2]]
if len(dec) < n + 2:
  sys.exit(f"only {len(dec)} decoder launches")
a, b = dec[n], dec[n + 1]
t_end = rows[a][1]
print(f"decoder launch {n}: {(rows[a][1] - rows[a][0]) / 1e3:.1f} us; to the start of launch {n + 1}: {(rows[b][0] - t_end) / 1e3:.1f} us")
busy = 0
for s, e, name in rows[a + 1:b]:
  gap = (s - t_end) / 1e3
  print(f"  {'+%.1f us idle' % gap if gap > 2 else '':>16}  {(e - s) / 1e3:9.1f} us  {name[:90]}")
  busy += e - s
  t_end = max(t_end, e)
gap = (rows[b][0] - t_end) / 1e3
print(f"  {'+%.1f us idle' % gap:>16}  next decoder launch ({(rows[b][1] - rows[b][0]) / 1e3:.1f} us)")
print(f"kernels and copies in between: {busy / 1e3:.1f} us (copies may overlap kernels)")
