#!/bin/bash
# Development aid (GPU box): per-kernel times (no output check: timing-aid builds included) of every zstandard_amd/lib/var_*.so
for f in zstandard_amd/lib/var_*.so; do
    echo -n "$f "; ZSMI_LIB_FILE=$PWD/$f timeout -k 10 200 python tools/time_kernels.py 2>&1 | tail -1 || exit 1
done
