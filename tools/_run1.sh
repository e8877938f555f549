timeout -k 10 200 python tools/gpu_debug.py > gpurun_out/dbg.log 2>&1; tail -12 gpurun_out/dbg.log
for a in "" "--chunks 2048 --chunk-size 131072"; do timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ratio'], j['roofline']['kernels_ms_per_step'])"; done
