#!/bin/bash
# PMC passes over the 256x128 LDS-DMA ping-pong kernel (tools/gemm_pmc.py: eager launches of its step shapes): round 4 = SQ (LDS pipe, MFMA,
# waits), round 5 = the texture path that carries the LDS-DMA stream (TA / TD / TCP busy + stall counters, LDS FIFO counters).
# (Round 5 also tried the TA_* / TD_* / TCP_* busy and stall counters of the texture path: those two passes did NOT finish inside 300 s each on this pool -- 10 GPU-minutes
# for nothing -- and are left out; `rocprofv3 --list-avail` names them.)  Counters only with --kernel-trace (the pool refuses PMC + other trace domains).      bash tools/pmc_big_gemm.sh [outdir]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo; export TMPDIR=/tmp
out=${1:-gpurun_out/r5_pmc_big}; mkdir -p $out
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/p$i -o pmc -- python tools/gemm_pmc.py > $out/p$i.log 2>&1; echo "pass $i rc=$?"
  db=$(find $out/p$i -name "*.db" | head -1); python tools/pmc_dump.py $db gemm_big > $out/p$i.txt 2>&1; cat $out/p$i.txt | head -20
done
find $out -name "*.db" -delete
