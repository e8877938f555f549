#!/bin/bash
# Development aid (GPU box): per-kernel compress times of every variant library on the classes named in CLASSES (default: zipf)
for f in zstandard_amd/lib/var_*.so; do
    for c in ${CLASSES:-zipf}; do
        echo -n "$f " ; CLS=$c ZSMI_LIB_FILE=$PWD/$f timeout -k 10 200 python tools/time_kernels.py 2>/dev/null | tail -1 || exit 1
    done
done
