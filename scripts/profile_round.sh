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
cd "$root"
python3 scripts/profile_summary.py "$tag"
