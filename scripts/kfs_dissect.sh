#!/bin/bash
# development: the streaming bottleneck-space kernels (csrc/tile_stream.hip) taken apart -- per-launch durations of the working-tree library and of
# the -DKFS_DISSECT / -DKFS_AUX / -DKFS_EXACT variant libraries (scripts/variant_lib.sh kfs_<name> tile_stream.hip -D...), same box
#   scripts/kfs_dissect.sh "<grep pattern>" name1 name2 ...
PAT=$1; shift
echo "== work"; scripts/fam_one.sh work "$PAT" | grep -v "^ms_per_step"
for n in "$@"; do
  echo "== $n"; AVMOE_LIB=$PWD/avmoe_amd/lib/variants/libkfs_$n.so scripts/fam_one.sh $n "$PAT" | grep -v "^ms_per_step\|^GPU"
done
