#!/bin/bash
# The compute-unit co-residency probe (scripts/mfma_probe.hip), every mode / mitigation, output kept in profiles/r05_mfma_probe.txt.
# Build first (in the container, hipcc cross-compiles):
#   for m in 0 1 2 3 4 5 8; do hipcc --offload-arch=gfx950 -O3 -DMIT=$m scripts/mfma_probe.hip -o avmoe_amd/lib/variants/probe/mfma_probe_mit$m; done
# then on the GPU box:  gpurun -- bash scripts/run_mfma_probe.sh
P=avmoe_amd/lib/variants/probe; O=gpurun_out/r5; mkdir -p $O; L=$O/mfma_probe.txt; : > $L
run() { echo "--- $*" >> $L; timeout 120 "$@" 2>&1 | grep -v "amdgpu.ids" >> $L; }
if [ "${1:-1}" = 1 ]; then
echo "## aggressor modes, victim as built (MIT 0: ds_read_b128 A operand, 9 KB static LDS, up to 8 blocks per CU)" >> $L
for mode in 0 1 2 3 4 5 6 7; do run $P/mfma_probe_mit0 6 $mode; done
echo "## mitigations in the victim's instruction stream, aggressor mode 3 (MFMA + ds_read_b128)" >> $L
for m in 1 2 3 4 5; do run $P/mfma_probe_mit$m 6 3; done
echo "## the A operand as two ds_read_b64 instead of one ds_read_b128 (MIT 8), aggressor modes 3 and 6" >> $L
run $P/mfma_probe_mit8 6 3; run $P/mfma_probe_mit8 6 6; run $P/mfma_probe_mit8 6 0
echo "## own blocks per CU beside the aggressor (mode 3): dynamic LDS request of the victim; 160 KB per CU, aggressor block = 16 KB" >> $L
echo "## (a) requests that leave room for aggressor blocks" >> $L
for lds in 20480 51200 92160; do run $P/mfma_probe_mit0 6 3 $lds; done
echo "## (b) requests with which N own blocks fill the CU's LDS (no 16 KB block fits beside them): N = 4, 2, 1" >> $L
for lds in 28672 67584 143360; do run $P/mfma_probe_mit0 6 3 $lds; done
echo "## (b) again with aggressor mode 6 (16x16x32 bf16 MFMA + ds_read_b128)" >> $L
for lds in 0 28672 67584 143360; do run $P/mfma_probe_mit0 6 6 $lds; done
cat $L
exit 0
fi
# ---- part 2 (scripts/run_mfma_probe.sh 2) (second call of the round): what in the aggressor matters, and is it the compute unit or the chip? ----
L2=$O/mfma_probe_part2.txt; : > $L2; L=$L2
echo "## aggressors with GEMM-like register use: sixteen independent chains of v_mfma_f32_16x16x32_bf16 (mode 8), + ds_read_b128 (mode 9)" >> $L
run $P/mfma_probe_mit0 6 8; run $P/mfma_probe_mit0 6 9
echo "## durations alone: victim only (mode 0), aggressor only (0 victim launches)" >> $L
run $P/mfma_probe_mit0 3 0; run $P/mfma_probe_mit0 3 1 0 1024 0; run $P/mfma_probe_mit0 3 1 0 128 0
echo "## compute unit or chip?  aggressor (mode 1, MFMA only) on 128 blocks = at most half of the 256 CUs, victim with a CU-filling LDS request (runs on CUs without an aggressor block, at the same time: see the duration)" >> $L
run $P/mfma_probe_mit0 6 1 143360 128; run $P/mfma_probe_mit0 6 1 0 128
run $P/mfma_probe_mit0 6 3 143360 128; run $P/mfma_probe_mit0 6 3 0 128
cat $L
