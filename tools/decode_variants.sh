for f in zstandard_amd/lib/libzsmi.so zstandard_amd/lib/var_*.so; do echo $f; ZSMI_LIB_FILE=$PWD/$f timeout -k 10 200 python tools/bench_decode.py 2>/dev/null | tail -1 | cut -c1-600 || exit 1; done
