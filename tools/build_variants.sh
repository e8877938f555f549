#!/bin/bash
# Development aid: builds kernel-shape variants of the library as zstandard_amd/lib/var_<name>.so (they travel to the GPU box);
# usage: tools/build_variants.sh name1:"-DX=1 -DY=2" name2:"..."   (tools/variants.sh benches them there)
rm -f zstandard_amd/lib/var_*.so
for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $flags -o zstandard_amd/lib/var_${name}.so zstandard_amd/csrc/zsmi_api.hip 2>&1 | grep -v 'warning: argument unused' &
    if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls -la zstandard_amd/lib/var_*.so
