#!/bin/bash
# Development aid (GPU box): PMC counters of the compress kernels of a debug-hook library at a stage stop of the sequences kernel
# usage: tools/pmc_stage.sh <tag> <lib.so> <ZSMI_STOP_SEQ value>   -> gpurun_out/pmc_<tag>_summary.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; lib=$2; stop=$3
export ZSMI_LIB_FILE=$PWD/$lib ZSMI_DEBUG_LIB=1 ZSMI_STOP_SEQ=$stop
run() { timeout -k 5 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$1 -- python tools/time_kernels.py > gpurun_out/pmc_${tag}_$1.log 2>&1 || exit 1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"
python tools/pmc_summary.py $tag | grep -E "kernel|k_encode_sequences"
