#!/bin/bash
# Builds the product library and the stage-stamp diagnostic build of the same sources.
set -e
cd "$(dirname "$0")/../tacotron_amd/csrc"
make -j4 2>&1 | grep -E "error|warning: (?!.*unused)" || true
mkdir -p ../../build_diag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTACO_STAMPS -DTACO_BT_STAMPS $TACO_EXTRA -shared -o ../../build_diag/libtaco_hip_stamps.so taco_api.hip taco_decoder.hip taco_decoder_batch.hip taco_layers.hip
ls -la ../libtaco_hip.so ../../build_diag/libtaco_hip_stamps.so
