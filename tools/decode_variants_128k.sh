for f in zstandard_amd/lib/libzsmi.so zstandard_amd/lib/var_*.so; do
    ZSMI_LIB_FILE=$PWD/$f timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --decode-frames 16384 --decode-frame-size 131072 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['decode']; print('$f', d['value'], d['roofline']['kernels_ms_per_step'])" || exit 1
done
