#!/bin/bash
# Build tuning variants of libmipt_hip.so: tools/build_variants.sh "NAME:-DFLAG=.. -DFLAG=.." ...
mkdir -p build_variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Iinclude -Wno-unused-result -Wno-unused-value $flags -shared -o build_variants/lib_$name.so pbrt-v3-spectral_amd/csrc/device/*.hip &
done
wait
ls build_variants
