#!/bin/bash
# development aid: kernel time when the entropy kernels stop after stage k (output invalid; timing only)
for k in 0 1 2 3; do
  ZSMI_STOP_LIT=$k ZSMI_STOP_SEQ=$k timeout -k 10 600 python tools/bench_noverify.py 2>&1 | tail -1
done
