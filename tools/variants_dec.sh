#!/bin/bash
# Development aid (GPU box): decode parity tests with one variant library, then the decode kernel times of every variant (see tools/variants.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$1" ]; then
    ZSMI_LIB_FILE=$PWD/zstandard_amd/lib/var_$1.so timeout -k 10 600 python -m pytest tests/test_gpu_codec.py tests/test_gpu_fuzz.py -m gpu -x -q -k "(decode or fuzz or checksum or zeros or roundtrip) and not intended" > gpurun_out/checkdec_$1.log 2>&1 || { tail -30 gpurun_out/checkdec_$1.log; exit 1; }
    tail -2 gpurun_out/checkdec_$1.log
fi
for f in zstandard_amd/lib/var_*.so; do
    ZSMI_LIB_FILE=$PWD/$f timeout -k 5 300 python tools/bench_decode.py --times-only ${DEC_ARGS} 2>/dev/null | tail -1 || exit 1
done
