#!/bin/bash
# development aid: bench at several internal-stream counts
for L in "$@"; do
  ZSMI_LANES=$L timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > /tmp/b.json
  python - "$L" <<'PY'
import sys, json
d = json.loads(open('/tmp/b.json').read())
print("lanes", sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
PY
done
