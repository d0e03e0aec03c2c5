set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
for i in 1 2; do
TACO_HIP_LIB=build_diag/libtaco_prev.so timeout -k 10 60 python3 scripts/decode_once.py 500 8
timeout -k 10 60 python3 scripts/decode_once.py 500 8
TACO_SOFTMAX_SHIFT=1 timeout -k 10 60 python3 scripts/decode_once.py 500 8
done
TACO_HIP_LIB=build_diag/libtaco_prev.so timeout -k 10 60 python3 scripts/decode_once.py 500 8 400
timeout -k 10 60 python3 scripts/decode_once.py 500 8 400
TACO_HIP_LIB=build_diag/libtaco_hip_stamps.so timeout -k 10 60 python3 scripts/decode_once.py 200 2 2> gpurun_out/stamps.log
