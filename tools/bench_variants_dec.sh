#!/bin/bash
# development aid (GPU box): tools/bench_decode.py for prebuilt library variants zstandard_amd/lib/variants/<name>.so
for v in zstandard_amd/lib/variants/*.so; do
  cp "$v" zstandard_amd/lib/libzsmi.so
  echo "== $(basename $v) $(timeout -k 10 200 python tools/bench_decode.py 2>&1 | tail -2 | cut -c1-140 | tr "\n" " ")"
done
