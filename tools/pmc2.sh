#!/bin/bash
# development aid: instruction-mix PMC passes (separate runs, kernel-trace only) over a one-step bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline "${EXTRA[@]}" > gpurun_out/pmc_${tag}_$1.log 2>&1; }
EXTRA=("$@")
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS"
run c "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"
python tools/pmc_summary.py $tag
