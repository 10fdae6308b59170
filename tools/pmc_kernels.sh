#!/bin/bash
# rocprofv3 counter passes over tools/probe_kernels.py (separate passes: SQ has 8 slots, TCC 4; never together with traces)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_probe; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/p$i -- python $R/tools/probe_kernels.py > $O/p$i.log 2>&1
  db=$(find $O/p$i -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/pmc_dump.py $db contraction > $O/p$i.txt 2>&1
done
find $O -name "*.db" -delete; rm -rf $O/p?/
grep -c . $O/p*.txt
