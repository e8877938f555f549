#!/bin/bash
# Development aid (GPU box): encoder parity tests (HIP == oracle E) with a variant library, then the bench line of every variant.
# usage: tools/check_variant.sh var_name   (zstandard_amd/lib/var_<name>.so is tested; all var_*.so are benched)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$1" ]; then
    ZSMI_LIB_FILE=$PWD/zstandard_amd/lib/var_$1.so timeout -k 10 600 python -m pytest tests/test_gpu_codec.py -m gpu -x -q -k "encode or mixed or zeros or units or joined or tiny or one_shot_large" > gpurun_out/check_$1.log 2>&1 || { tail -30 gpurun_out/check_$1.log; exit 1; }
    tail -3 gpurun_out/check_$1.log
fi
bash tools/variants.sh
