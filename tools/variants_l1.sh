#!/bin/bash
# Development aid (GPU box): level-1 and 128 KiB bench lines of every variant library (see tools/variants.sh)
for f in zstandard_amd/lib/var_*.so; do
  for args in "--level 1" "--level 1 --chunks 2048 --chunk-size 131072" "--chunks 2048 --chunk-size 131072"; do
    ZSMI_LIB_FILE=$PWD/$f timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --decode-frames 0 $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$f', '$args', d['value'], d['ratio'], d['roofline']['kernels_ms_per_step'])" || exit 1
  done
done
