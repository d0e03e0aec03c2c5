#!/bin/bash
# Where the convolution kernel's LDS bank conflicts come from and what they cost: the product library against timing-only
# builds whose staging writes go to conflict-free (wrong) addresses (weights: ws[buf][0][(tid * 2 + 512 * it) % ...], inputs:
# xs[buf][0][(tid + 256 * item) % ...], two three-line edits of store_item in taco_layers.hip as it was before the inputs'
# row-per-instruction map; built by hand into build_diag/libtaco_cv_{ww,xw,both}.so).  Per build: the postnet's time at 38400 columns and one counter pass
# (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE).  Run from the repository root: bash scripts/conv_conflict_ab.sh <tag>
set -e
tag=${1:?tag}
root=$(pwd)
out=$root/gpurun_out/prof_${tag}_conv_ab
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for v in product ww xw both; do
  if [ $v = product ]; then export TACO_HIP_LIB=$root/tacotron_amd/libtaco_hip.so; else export TACO_HIP_LIB=$root/build_diag/libtaco_cv_$v.so; fi
  timeout -k 10 100 python3 "$root/scripts/postnet_once.py" 38400 > "$out/time_$v.txt" 2>/dev/null
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d "$out/pmc_$v" -o pmc -- python3 "$root/scripts/postnet_once.py" 38400 > "$out/pmc_$v.log" 2>&1
done
cd "$root"
python3 - "$out" "$tag" <<'PY'
import json, sqlite3, sys
from pathlib import Path
out, tag = Path(sys.argv[1]), sys.argv[2]
res = {}
for v in ("product", "ww", "xw", "both"):
  dbs = sorted((out / f"pmc_{v}").rglob("*_results.db"))
  c = {}
  if dbs:
    con = sqlite3.connect(dbs[0])
    q = ("select counter_name, sum(value) from counters_collection where kernel_name like '%taco_conv_kernel<5%' group by counter_name")
    c = dict(con.execute(q).fetchall())
  t = (out / f"time_{v}.txt").read_text().strip()
  res[v] = {"timing": t, "counters": c,
            "lds_bank_conflict_cycles_per_active_cycle": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4) if c.get("SQ_LDS_IDX_ACTIVE") else None}
res["builds"] = {"product": "the library as built", "ww": "weight staging writes to conflict-free addresses (results wrong: timing only)",
                 "xw": "input staging writes to conflict-free addresses (timing only)", "both": "both"}
json.dump(res, open(f"profiles/{tag}_conv_conflicts.json", "w"), indent=1)
print(json.dumps({k: (v["timing"], v["lds_bank_conflict_cycles_per_active_cycle"]) for k, v in res.items() if k != "builds"}, indent=1))
PY
