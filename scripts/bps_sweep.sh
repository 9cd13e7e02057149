#!/bin/bash
# development: tile-kernel time vs blocks per sample (AVMOE_BPS="<audio>,<visual>")
for b in "4,4" "4,1" "4,2" "4,3" "2,2" "8,2" "16,4"; do
  echo "== AVMOE_BPS=$b"
  AVMOE_BPS=$b python scripts/prof_shapes.py 2>&1 | grep -E "^total|k_(pre|post|mid)" 
done
