#!/bin/bash
# Same-box timing of several builds over input lengths: bash scripts/ab_lens.sh "<len> ..." <lib> [<lib> ...]
set -e
lens=$1; shift
for n in $lens; do
  for lib in "$@"; do
    TACO_HIP_LIB=$lib timeout -k 10 60 python3 scripts/decode_once.py 500 6 $n
  done
done
