#!/bin/bash
# development aid: instruction-mix counters of the decode kernel (separate PMC passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_dec_$1 -- python tools/bench_decode.py --steps 1 --warmup 1 > gpurun_out/pmc_dec_$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run c "SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS"
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(float); cnt = collections.Counter()
for d in "abc":
    for f in glob.glob(f"gpurun_out/pmc_dec_{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_decode_frames" in r["Kernel_Name"]:
                res[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(res): print(f"{k:28s} {res[k]/cnt[k]:16.0f}")
PY
