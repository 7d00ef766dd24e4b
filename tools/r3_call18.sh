#!/bin/bash
out=gpurun_out/r3_c18; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
PH_ATTN_QT2=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OLDPWD/$out/p1 -o p -- python $OLDPWD/tools/attn_pmc.py > $OLDPWD/$out/p1.log 2>&1
PH_ATTN_QT2=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OLDPWD/$out/p2 -o p -- python $OLDPWD/tools/attn_pmc.py > $OLDPWD/$out/p2.log 2>&1
cd $OLDPWD
for d in p1 p2; do
DB=$(find $out/$d -name "*.db" | head -1)
python tools/pmc_dump.py $DB attn > $out/$d.txt 2>&1; cat $out/$d.txt
done
tail -3 $out/p1.log
find $out -name "*.db" -size +20M -delete
cd /tmp
PH_ATTN_QT2=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT -d $OLDPWD/gpurun_out/r3_c18/p3 -o p -- python $OLDPWD/tools/attn_pmc.py > $OLDPWD/gpurun_out/r3_c18/p3.log 2>&1
cd $OLDPWD
DB=$(find gpurun_out/r3_c18/p3 -name "*.db" | head -1)
python tools/pmc_dump.py $DB attn > gpurun_out/r3_c18/p3.txt 2>&1; cat gpurun_out/r3_c18/p3.txt; tail -3 gpurun_out/r3_c18/p3.log
find gpurun_out/r3_c18 -name "*.db" -size +20M -delete
