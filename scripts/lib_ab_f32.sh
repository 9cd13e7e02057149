#!/bin/bash
# dev: fp32 cfg-2 engine products and step under variant libraries      scripts/lib_ab_f32.sh "base sgb3 sgb5" [grep pattern]
for v in $1; do
  if [ $v = base ]; then unset AVMOE_LIB; else export AVMOE_LIB=$PWD/avmoe_amd/lib/variants/lib$v.so; fi
  echo "== $v"
  FAM_OUT=gpurun_out/r6/ab scripts/fam_one.sh $v "${2:-gemm}" --dtype f32 2>/dev/null | head -${3:-12}
done
