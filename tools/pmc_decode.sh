#!/bin/bash
# development aid: instruction-mix and memory-path counters of the decode kernels (separate PMC passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "pass $1"; timeout -k 5 150 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_dec_$1 -- python tools/bench_decode.py --steps 1 --warmup 1 > gpurun_out/pmc_dec_$1.log 2>&1 || echo "pass $1 failed"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run c "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
run d "TA_TA_BUSY TA_TOTAL_WAVEFRONTS"
run e "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ"
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for d in "abcde":
    for f in glob.glob(f"gpurun_out/pmc_dec_{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("k_dec"):
                res[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(res):
    print(k)
    for c in sorted(res[k]): print(f"   {c:28s} {res[k][c]/cnt[k][c]:16.0f}")
PY
