#!/usr/bin/env python3
"""Per-kernel calls / total / average / min / max of one `rocprofv3 --kernel-trace` output directory as a csv.
Usage: kernel_stats.py <rocprofv3 output dir> <out.csv>"""
import csv, sqlite3, sys
from pathlib import Path

found = sorted(Path(sys.argv[1]).rglob("*_results.db"))
if not found:
  sys.exit(f"no database under {sys.argv[1]}")
con = sqlite3.connect(found[0])
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                   "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
  w = csv.writer(f)
  w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
  for name, calls, tot, avg, lo, hi in rows:
    w.writerow([name[:100], calls, tot, round(avg, 1), round(100.0 * tot / total, 2), lo, hi])
