#!/bin/bash
# development: build a variant library with one translation unit recompiled under extra flags
#   scripts/variant_lib.sh <name> <source under csrc> <extra hipcc flags...>   ->  avmoe_amd/lib/variants/lib<name>.so   (use with AVMOE_LIB=)
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/avmoe_amd/lib/variants
obj=$root/avmoe_amd/lib/variants/$name.$src.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $root/avmoe_amd/csrc/$src -o $obj
others=$(ls $root/avmoe_amd/lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $obj $others -o $root/avmoe_amd/lib/variants/lib$name.so
echo $root/avmoe_amd/lib/variants/lib$name.so
