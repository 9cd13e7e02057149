#!/bin/bash
# development: the whole library built with -DAVMOE_DEV (csrc/common.h: dev_env -- the A/B switches of scripts/README.md compiled in)
#   scripts/dev_lib.sh [extra hipcc flags]  ->  avmoe_amd/lib/variants/libdev.so   (use with AVMOE_LIB=$PWD/avmoe_amd/lib/variants/libdev.so)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/avmoe_amd/lib/variants/dev
mkdir -p $O
n=0
for f in $R/avmoe_amd/csrc/*.hip $R/avmoe_amd/csrc/*.cpp; do
  case $(basename $f) in host_*) continue;; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DAVMOE_DEV "$@" -c $f -o $O/$(basename $f).o &
  n=$((n+1)); if [ $((n % 8)) -eq 0 ]; then wait; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o $R/avmoe_amd/lib/variants/libdev.so && echo $R/avmoe_amd/lib/variants/libdev.so
