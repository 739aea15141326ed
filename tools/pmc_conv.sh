#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_conv
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcc_$i -o pmc -- python tools/pmc_conv.py > gpurun_out/pmc_conv/pass$i.log 2>&1 < /dev/null
  echo "pass $i rc=$?"
done
PMC_NAME_LEN=70 python tools/pmc_summarize.py /tmp/pmcc_1 /tmp/pmcc_2 /tmp/pmcc_3 | grep -E "igemm|wgrad|wino" | tee gpurun_out/pmc_conv/summary${SPI_BENCH_F16:-0}.txt
