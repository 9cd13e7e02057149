#!/bin/bash
# Round 6, second probe call: is it the DEPENDENT matrix instruction (accumulator chained from the previous one) that goes wrong beside another
# kernel's long matrix instructions?  MIT 22 / 23: the split-bf16 / exact-fp32 mat-vec with every product into a ZERO accumulator, summed by the VALU.
P=avmoe_amd/lib/variants/probe; O=gpurun_out/r6; mkdir -p $O; L=$O/mfma_probe4.txt; : > $L
run() { echo "--- $*" >> $L; timeout 300 "$@" 2>&1 | grep -v "amdgpu.ids" | grep -v "^rep \|^last rep" >> $L; }
for m in 22 23; do for mode in 0 1 3 6 7 8 9; do run $P/mfma_probe_mit$m 100 $mode; done; done
cat $L
