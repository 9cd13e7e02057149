#!/bin/bash
# development: tile-kernel families under each variant library in avmoe_amd/lib/variants/ (plus the default build)
pat=${1:-"^total|k_(pre|post|mid)"}
echo "== default"; python scripts/prof_shapes.py 2>&1 | grep -E "$pat"
for l in avmoe_amd/lib/variants/lib*.so; do
  echo "== $l"; AVMOE_LIB=$PWD/$l python scripts/prof_shapes.py 2>&1 | grep -E "$pat"
done
