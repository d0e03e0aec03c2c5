#!/bin/bash
# One measurement pass on the GPU box: bench line, kernel trace of the same bench command, and the
# two HBM traffic counters of the decoder in passes of their own.  Run from the repository root:
#   bash scripts/profile_round.sh r1g
# Everything lands under gpurun_out/prof_<tag>/; scripts/profile_summary.py turns the databases
# into the small files kept under profiles/.
set -e
tag=${1:?tag}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 240 python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- python3 "$root/bench.py" --cpu-seconds 0 > "$out/bench_under_prof.json" 2> "$out/kt.err"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch -- python3 "$root/scripts/decode_once.py" 1000 2 > "$out/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write -- python3 "$root/scripts/decode_once.py" 1000 2 > "$out/write.log" 2>&1
# where the decoder's wave cycles go (issue, waits) and what its LDS traffic looks like: two more counter passes
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d "$out/sqa" -o sqa -- python3 "$root/scripts/decode_once.py" 1000 2 > "$out/sqa.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 -d "$out/sqb" -o sqb -- python3 "$root/scripts/decode_once.py" 1000 2 > "$out/sqb.log" 2>&1
cd "$root"
python3 scripts/profile_summary.py "$tag"
