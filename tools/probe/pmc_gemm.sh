#!/bin/bash
# SQ counters of one GEMM shape (tools/gemm_sweep.py spec) -> gpurun_out/<tag>/pmc_*.txt
spec=$1; tag=${2:-pmcg}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p -- python3 tools/gemm_sweep.py $spec > $out/p$i.log 2>&1
  python3 tools/pmc_summary.py $out/p$i/p_counter_collection.csv 3 > $out/pmc_$i.txt 2>&1
  rm -rf $out/p$i
done
cat $out/pmc_*.txt | grep -v "^TOTAL"
