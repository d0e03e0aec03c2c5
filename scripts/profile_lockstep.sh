#!/bin/bash
# One measurement pass of the lockstep decoder on the GPU box: a batch of 64 lines, 600 frames each
# (scripts/batch_once.py 64 600), traced, then its matrix-core, issue and traffic counters in passes of
# their own.  Run from the repository root:  bash scripts/profile_lockstep.sh r2b
# Results under gpurun_out/prof_<tag>/; scripts/profile_lockstep_summary.py writes the files kept under profiles/.
set -e
tag=${1:?tag}
root=$(pwd)
out=$root/gpurun_out/prof_${tag}_lockstep
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 200 python3 scripts/batch_once.py 64 600 3 > "$out/batch64.txt" 2> "$out/batch64.err"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- python3 "$root/scripts/batch_once.py" 64 600 3 > "$out/kt.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d "$out/sqa" -o sqa -- python3 "$root/scripts/batch_once.py" 64 600 1 > "$out/sqa.log" 2>&1
# (TACO_PRECISION=bf16 in the environment profiles the bf16 setting: its gate products are bf16 matrix-core instructions)
if [ "$TACO_PRECISION" = bf16 ]; then mfma="SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_MOPS_BF16"; else mfma="SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES"; fi
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA $mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -d "$out/sqm" -o sqm -- python3 "$root/scripts/batch_once.py" 64 600 1 > "$out/sqm.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d "$out/sqb" -o sqb -- python3 "$root/scripts/batch_once.py" 64 600 1 > "$out/sqb.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch -- python3 "$root/scripts/batch_once.py" 64 600 1 > "$out/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write -- python3 "$root/scripts/batch_once.py" 64 600 1 > "$out/write.log" 2>&1
cd "$root"
python3 scripts/profile_lockstep_summary.py "$tag"
