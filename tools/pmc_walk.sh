#!/bin/bash
# development aid: PMC passes for the LZ kernels (separate runs, kernel-trace only); usage: tools/pmc_walk.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --decode-frames 0 > gpurun_out/pmc_${tag}_$1.log 2>&1 || exit 1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
run b "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
run e "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
run f "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"
python tools/pmc_summary.py $tag
