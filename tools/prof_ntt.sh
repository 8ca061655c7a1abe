#!/bin/bash
# tools/prof_ntt.sh : kernel trace + counter passes of the 2^22 NTT (tools/bench_ntt.py); counters in their own passes
OUT=gpurun_out/ntt_prof; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/bench_ntt.py --sizes 22 --steps 4"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- $CMD > $R/$OUT/trace.log 2>&1); echo "trace rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/pmc_a -- $CMD > $R/$OUT/pmc_a.log 2>&1); echo "pmc_a rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/$OUT/pmc_b -- $CMD > $R/$OUT/pmc_b.log 2>&1); echo "pmc_b rc=$?"
(cd /tmp && timeout 120 rocprofv3 -L > $R/$OUT/counters.txt 2>&1); echo "list rc=$?"
python $R/tools/pmc_summary.py $OUT/pmc_a $OUT/pmc_b --out $OUT/pmc.json 2>&1 | tail -2
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
ls $OUT/trace/*/ 2>/dev/null | head
