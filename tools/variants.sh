#!/bin/bash
# Development aid (GPU box): bench line of every kernel-shape variant built as zstandard_amd/lib/var_<name>.so
# (hipcc ... -D<shape macro> -o zstandard_amd/lib/var_X.so zstandard_amd/csrc/zsmi_api.hip); prints value and per-kernel times.
for f in zstandard_amd/lib/var_*.so; do
    ZSMI_LIB_FILE=$PWD/$f timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$f', d['value'], d['ratio'], d['roofline']['kernels_ms_per_step'])" || exit 1
done
