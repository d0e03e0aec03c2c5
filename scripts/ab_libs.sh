#!/bin/bash
# Same-box timing of several builds of the library: bash scripts/ab_libs.sh <frames> <lib> [<lib> ...]
# (each twice, interleaved, so drift of the box shows).
set -e
frames=$1; shift
for i in 1 2; do
  for lib in "$@"; do
    TACO_HIP_LIB=$lib timeout -k 10 60 python3 scripts/decode_once.py "$frames" 8
  done
done
