#!/bin/bash
# development: build avmoe_amd/lib/variants/libhead.so from the COMMITTED sources (git HEAD, or the revision given as $1) -- the "before" side of a same-box
# A/B against the working tree:   scripts/head_lib.sh && gpurun -- 'scripts/ab_bench.sh - "AVMOE_LIB=$PWD/avmoe_amd/lib/variants/libhead.so"'
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/oldsrc
mkdir -p $O $R/avmoe_amd/lib/variants && git -C $R archive ${1:-HEAD} avmoe_amd/csrc include | tar -x -C $O
for f in $O/avmoe_amd/csrc/*.hip $O/avmoe_amd/csrc/*.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c $f -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/avmoe_amd/csrc/*.o -o $R/avmoe_amd/lib/variants/libhead.so && echo $R/avmoe_amd/lib/variants/libhead.so
