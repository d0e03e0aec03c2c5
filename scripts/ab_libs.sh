#!/bin/bash
# Same-box timing of several builds of the library on the batch of 64: ab_libs.sh name1 name2 ... (build_diag/libtaco_<name>.so;
# "base" = the in-tree library).  Two passes, so that drift of the box shows.
for pass in 1 2; do
  for n in "$@"; do
    if [ "$n" = base ]; then lib=tacotron_amd/libtaco_hip.so; else lib=build_diag/libtaco_$n.so; fi
    echo -n "$n: "
    TACO_HIP_LIB=$lib timeout -k 10 100 python3 scripts/batch_once.py 64 600 3 2>/dev/null | tail -n 1 || exit 1
  done
done
