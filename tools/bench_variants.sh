#!/bin/bash
# development aid (GPU box): run bench.py for several prebuilt library variants named zstandard_amd/lib/variants/<name>.so
for v in zstandard_amd/lib/variants/*.so; do
  cp "$v" zstandard_amd/lib/libzsmi.so
  echo "== $(basename $v)"
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ratio'], j['roofline']['kernels_ms_per_step'])"
done
