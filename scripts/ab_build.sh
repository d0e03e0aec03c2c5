#!/bin/bash
# Builds the committed sources as build_diag/libtaco_prev.so and the working tree as the product
# library + the stage-stamp diagnostic build, for same-box A/B timing with scripts/decode_once.py.
set -e
cd "$(dirname "$0")/.."
git stash -q
bash scripts/build_all.sh >/dev/null 2>&1 || { git stash pop -q; exit 1; }
cp tacotron_amd/libtaco_hip.so build_diag/libtaco_prev.so
git stash pop -q
bash scripts/build_all.sh | tail -1
