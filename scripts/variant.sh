#!/bin/bash
# Builds a variant of the lockstep decoder for same-box A/B timing: scripts/variant.sh NAME [compiler flags ...]
# -> build_diag/libtaco_NAME.so (taco_decoder_batch.hip compiled with the flags, the other objects as built in-tree).
set -e
name=${1:?name}; shift
cd "$(dirname "$0")/../tacotron_amd/csrc"
make -j4 >/dev/null 2>&1
mkdir -p ../../build_diag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c taco_decoder_batch.hip -o /tmp/tdb_$name.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_diag/libtaco_$name.so taco_api.o taco_decoder.o taco_layers.o /tmp/tdb_$name.o
ls -la ../../build_diag/libtaco_$name.so
