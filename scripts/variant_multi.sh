#!/bin/bash
# development: a variant library with SEVERAL translation units recompiled under the same extra flags
#   scripts/variant_multi.sh <name> "a.hip b.hip ..." <extra hipcc flags...>   ->  avmoe_amd/lib/variants/lib<name>.so   (use with AVMOE_LIB=)
set -e
name=$1; srcs=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/avmoe_amd/lib/variants
objs=""; skip=""
for src in $srcs; do
  obj=$root/avmoe_amd/lib/variants/$name.$src.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $root/avmoe_amd/csrc/$src -o $obj &
  objs="$objs $obj"; skip="$skip -e /$src.o"
done
wait
others=$(ls $root/avmoe_amd/lib/obj/*.o | grep -v $skip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $others -o $root/avmoe_amd/lib/variants/lib$name.so
rm -f $objs
echo $root/avmoe_amd/lib/variants/lib$name.so
