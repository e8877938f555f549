#!/bin/bash
# development aid: vector-memory path PMC passes (TA / TCP: two counters of a block per pass) over a one-step bench.
# usage: bash tools/pmc_mem.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
EXTRA=("$@")
run() { echo "pass $1: $2"; timeout -k 5 150 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline "${EXTRA[@]}" > gpurun_out/pmc_${tag}_$1.log 2>&1 || echo "pass $1 failed"; }
run a "TA_TA_BUSY TA_TOTAL_WAVEFRONTS"
run b "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
run c "TCP_GATE_EN1 TCP_GATE_EN2"
run d "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ"
run e "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES"
run f "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES"
python tools/pmc_summary.py $tag
