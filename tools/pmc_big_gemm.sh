cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
out=gpurun_out/r4_pmc_big; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $out/p1 -o pmc -- python tools/gemm_pmc.py > $out/p1.log 2>&1; echo rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $out/p2 -o pmc -- python tools/gemm_pmc.py > $out/p2.log 2>&1; echo rc=$?
for p in p1 p2; do db=$(find $out/$p -name "*.db" | head -1); python tools/pmc_dump.py $db gemm_big > $out/$p.txt 2>&1; cat $out/$p.txt | head -40; done
tail -3 $out/p1.log
find $out -name "*.db" -delete
