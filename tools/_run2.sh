for b in 8192 2048 1024 512 256; do echo "== blocks in flight $b"; ZSMI_BLOCKS_IN_FLIGHT=$b timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ratio'], j['roofline']['kernels_ms_per_step'])"; done
