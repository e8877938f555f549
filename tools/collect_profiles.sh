#!/bin/bash
# GPU box: everything bench.py's roofline blocks cite, in one go.  usage: bash tools/collect_profiles.sh <tag>   (copy gpurun_out/<tag>_* to profiles/)
#  - <tag>_bench.json                  plain bench.py line
#  - <tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of the compress workload alone
#  - <tag>_pmc_summary.csv             PMC passes (each its own run, kernel-trace only), compress kernels
#  - <tag>_traffic.json                HBM bytes per launch from FETCH_SIZE / WRITE_SIZE, compress kernels
#  - <tag>_decode_kernel_stats.csv, <tag>_decode_pmc_summary.csv, <tag>_decode_traffic.json   the same for the decode workload (tools/bench_decode.py)
#  - <tag>_bench_128k.json, <tag>_bench_128k_l1.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout -k 5 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- python bench.py --no-cpu-baseline --no-extras --decode-frames 0 > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_rocprof.err || exit 1
cp $(ls gpurun_out/prof_${tag}_stats/*/*_kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats.csv
run() { echo "pmc pass $1"; timeout -k 5 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --decode-frames 0 > gpurun_out/pmc_${tag}_$1.log 2>&1 || exit 1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"
run c "FETCH_SIZE"
run d "WRITE_SIZE"
run e "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_WR"
python tools/pmc_summary.py $tag
cp gpurun_out/pmc_${tag}_summary.csv gpurun_out/${tag}_pmc_summary.csv
drun() { echo "decode pmc pass $1"; timeout -k 5 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}dec_$1 -- python tools/bench_decode.py --steps 1 --warmup 1 > gpurun_out/pmc_${tag}dec_$1.log 2>&1 || exit 1; }
drun a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
drun b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
drun c "FETCH_SIZE"
drun d "WRITE_SIZE"
drun e "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
python tools/pmc_summary.py ${tag}dec
cp gpurun_out/pmc_${tag}dec_summary.csv gpurun_out/${tag}_decode_pmc_summary.csv
python - "$tag" <<'PY'
import csv, json, sys, os, re
sys.path.insert(0, "tools")
from src_fingerprint import fingerprint
tag = sys.argv[1]
fp = fingerprint(os.getcwd())
NOTE = ("bytes per launch = counter * 1024.  On gfx950 FETCH_SIZE tallies a wide coalesced 16 B/lane stream at half its bytes (MI355X guide, HBM section): "
        "the kernels that read the SOURCE that way get half the source bytes added back, as the guide prescribes - k_lz_walk* (staging: the source once), "
        "k_encode_literals (the literal compaction reads the block 16 B a lane), k_dec_execute (literals 16 B a lane: the literal bytes are not known per launch here, so the "
        "compressed input is taken as the lower bound of what was under-read and NOT added: its FETCH is given raw and marked); every other load is <= 8 B a lane or a gather and is given raw.")
def table(sumfile, blocks, fix):
    out = {}
    for r in csv.DictReader(open(sumfile)):
        name = re.sub(r"(_\d+|_true|_false)+$", "", r["kernel"])
        f = int(r.get("FETCH_SIZE", 0)) * 1024; w = int(r.get("WRITE_SIZE", 0)) * 1024
        add = fix(name)
        e = out.setdefault(name, {"fetch_bytes": 0, "write_bytes": 0, "hbm_bytes": 0, "fetch_correction_bytes": 0})
        e["fetch_bytes"] += f + add; e["write_bytes"] += w; e["hbm_bytes"] += f + add + w; e["fetch_correction_bytes"] += add
        e.update(blocks)
    return out
S = 4096 * 65536
comp = table(f"gpurun_out/pmc_{tag}_summary.csv", {"blocks_per_launch": 4096}, lambda n: S // 2 if (n.startswith("k_lz_walk") or n == "k_encode_literals") else 0)
json.dump({"kernel_source_sha256": fp, "source": f"profiles/{tag}_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace only; bench.py --steps 1 --warmup 1: 4096 x 64 KiB chunks, level 3)",
           "note": NOTE, "kernels": comp}, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
dec = table(f"gpurun_out/pmc_{tag}dec_summary.csv", {"frames_per_launch": 57344}, lambda n: 0)
json.dump({"kernel_source_sha256": fp, "source": f"profiles/{tag}_decode_pmc_summary.csv (tools/bench_decode.py --steps 1 --warmup 1: 57344 frames of 32 KiB in one launch per kernel)",
           "note": NOTE, "kernels": dec}, open(f"gpurun_out/{tag}_decode_traffic.json", "w"), indent=1)
print(json.dumps({k: v["hbm_bytes"] for k, v in comp.items()})); print(json.dumps({k: v["hbm_bytes"] for k, v in dec.items()}))
PY
# decode side: kernel stats of tools/bench_decode.py and its JSON line
timeout -k 5 240 python tools/bench_decode.py > gpurun_out/${tag}_decode_bench.json 2> gpurun_out/${tag}_decode_bench.err || true
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_dec -- python tools/bench_decode.py > gpurun_out/${tag}_decode_under_rocprof.json 2> gpurun_out/${tag}_decode_rocprof.err || true
cp $(ls gpurun_out/prof_${tag}_dec/*/*_kernel_stats.csv | head -1) gpurun_out/${tag}_decode_kernel_stats.csv
# 128 KiB chunks and level 1 (BASELINE configs 3 and 5 shapes)
timeout -k 5 240 python bench.py --chunks 2048 --chunk-size 131072 --no-cpu-baseline --no-extras --decode-frames 0 > gpurun_out/${tag}_bench_128k.json 2>/dev/null || true
timeout -k 5 240 python bench.py --chunks 2048 --chunk-size 131072 --level 1 --no-cpu-baseline --no-extras --decode-frames 0 > gpurun_out/${tag}_bench_128k_l1.json 2>/dev/null || true
# last: the bench line once more with this run's traffic files in place (bench.py reports roofline.traffic only from files measured at the
# built library's sources), so that the committed line carries it
round=${tag%%_*}
cp gpurun_out/${tag}_traffic.json profiles/${round}_traffic.json && cp gpurun_out/${tag}_decode_traffic.json profiles/${round}_decode_traffic.json
timeout -k 5 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
