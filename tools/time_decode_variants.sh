#!/bin/bash
# Development aid (GPU box): per-kernel decode times (no output check: timing-aid builds included) of every zstandard_amd/lib/var_*.so
# usage: tools/time_decode_variants.sh [extra bench_decode.py flags, e.g. --libzstd]
for f in zstandard_amd/lib/var_*.so; do
    ZSMI_LIB_FILE=$PWD/$f timeout -k 10 200 python tools/bench_decode.py --times-only --steps 3 "$@" 2>&1 | tail -1 || exit 1
done
