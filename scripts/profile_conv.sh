#!/bin/bash
# Counter passes over the postnet's convolution kernel at a large column count (one utterance of T columns; 38400 is
# what a batch of 64 x 600 frames gives the kernel).  Run from the repository root:  bash scripts/profile_conv.sh r2h [T]
# Results under gpurun_out/prof_<tag>_conv/, summary printed by scripts/profile_conv_summary.py.
set -e
tag=${1:?tag}
T=${2:-38400}
root=$(pwd)
out=$root/gpurun_out/prof_${tag}_conv
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 120 python3 scripts/postnet_once.py "$T" > "$out/timing.txt" 2> "$out/timing.err"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- python3 "$root/scripts/postnet_once.py" "$T" > "$out/kt.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -d "$out/sqa" -o sqa -- python3 "$root/scripts/postnet_once.py" "$T" > "$out/sqa.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$out/sqm" -o sqm -- python3 "$root/scripts/postnet_once.py" "$T" > "$out/sqm.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_ANY -d "$out/sqb" -o sqb -- python3 "$root/scripts/postnet_once.py" "$T" > "$out/sqb.log" 2>&1
cd "$root"
python3 scripts/profile_conv_summary.py "$tag"
