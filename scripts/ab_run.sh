#!/bin/bash
# GPU tests of the working tree, then same-box timing against build_diag/libtaco_prev.so (the committed
# sources, built by scripts/ab_build.sh) and the stage stamps of the diagnostic build.
set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
for i in 1 2; do
  TACO_HIP_LIB=build_diag/libtaco_prev.so timeout -k 10 60 python3 scripts/decode_once.py 1000 8
  timeout -k 10 60 python3 scripts/decode_once.py 1000 8
done
for n in 30 200 256 400; do
  TACO_HIP_LIB=build_diag/libtaco_prev.so timeout -k 10 60 python3 scripts/decode_once.py 500 6 $n
  timeout -k 10 60 python3 scripts/decode_once.py 500 6 $n
done
TACO_HIP_LIB=build_diag/libtaco_prev.so timeout -k 10 60 python3 scripts/encode_once.py 50
timeout -k 10 60 python3 scripts/encode_once.py 50
TACO_HIP_LIB=build_diag/libtaco_hip_stamps.so timeout -k 10 60 python3 scripts/decode_once.py 200 2 2> gpurun_out/stamps.log
