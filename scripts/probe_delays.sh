#!/bin/bash
# Where the lockstep frame's critical path runs: builds of the lockstep decoder in which the workgroups of ONE role lose
# 16 naps of 64 clocks (about 0.47 us at 2.2 GHz) at ONE point of every frame (-DBT_PROBE_AT / -DBT_PROBE_ROLE, timing
# only: results are unchanged), each timed on the batch of 64 against the in-tree library.  A point on the critical path
# makes the frame longer by the whole delay, a point with slack by nothing.
# Points: 1 before the attention LSTM's products (prenet output loaded), 2 Q: before the query rows, 3 Q: before the
# softmax (partial energies gathered), 4 before the decoder LSTM's reduction (context products issued), 5 before the
# decoder hidden state is fetched (P / M: their rows follow), 6 the frame's end (Q: before the location term).
# Roles: 0 plain, 1 P, 2 M, 3 Q.     Usage (GPU box, repository root): bash scripts/probe_delays.sh build | run
set -e
cd "$(dirname "$0")/.."
variants="1:0 1:1 1:3 2:3 3:3 4:0 4:1 4:3 5:0 5:1 5:2 5:3 6:0 6:3"
if [ "$1" = build ]; then
  for v in $variants; do
    at=${v%:*}; role=${v#*:}
    bash scripts/variant.sh probe_${at}_${role} -DBT_PROBE_AT=$at -DBT_PROBE_ROLE=$role | tail -n 1
  done
  exit 0
fi
names=base
for v in $variants; do names="$names probe_${v%:*}_${v#*:}"; done
bash scripts/ab_libs.sh $names base
