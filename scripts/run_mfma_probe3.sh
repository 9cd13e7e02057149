#!/bin/bash
# Round 6: the co-residency probe with the victim in the form the TUNED kernels run (MIT 20: split-bf16 mat-vecs on v_mfma_f32_16x16x32_bf16 as
# csrc/tile_fast.hip::mmT_split / csrc/tile_stream.hip::mm_presplit; MIT 21: + their DPP column sums), 200 repetitions beside the aggressors
# that corrupt the fp32-MFMA form (MIT 0).  Build (container): for m in 0 20 21; do hipcc --offload-arch=gfx950 -O3 -DMIT=$m scripts/mfma_probe.hip
#   -o avmoe_amd/lib/variants/probe/mfma_probe_mit$m; done ; on the GPU box: bash scripts/run_mfma_probe3.sh -> gpurun_out/r6/mfma_probe3.txt
P=avmoe_amd/lib/variants/probe; O=gpurun_out/r6; mkdir -p $O; L=$O/mfma_probe3.txt; : > $L
run() { echo "--- $*" >> $L; timeout 300 "$@" 2>&1 | grep -v "amdgpu.ids" >> $L; }
echo "## control: the fp32-MFMA victim (MIT 0) alone and beside the dense aggressors (modes 3: 32x32x16 bf16 + ds_read_b128, 8: sixteen chains of 16x16x32 bf16)" >> $L
run $P/mfma_probe_mit0 20 0; run $P/mfma_probe_mit0 20 3; run $P/mfma_probe_mit0 20 8
echo "## the split-bf16 victim (MIT 20), 200 repetitions per aggressor" >> $L
for mode in 0 1 3 6 8 9; do run $P/mfma_probe_mit20 200 $mode; done
echo "## + DPP column sums (MIT 21), 200 repetitions" >> $L
for mode in 0 3 8 9; do run $P/mfma_probe_mit21 200 $mode; done
cat $L
