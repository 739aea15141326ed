#!/bin/bash
# SQ counters of one config-size render forward + backward (tools/pmc_render.py): where do the decoder-backward kernels' wave cycles go?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_scatter
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_$i -o pmc -- python tools/pmc_render.py > gpurun_out/pmc_scatter/pass$i.log 2>&1 < /dev/null
  echo "pass $i rc=$?"
done
python tools/pmc_summarize.py /tmp/pmcs_1 /tmp/pmcs_2 /tmp/pmcs_3 /tmp/pmcs_4 /tmp/pmcs_5 | grep -E "decode_bwd_tiled|plane_scatter|bin_points|raymarch_bwd" | tee gpurun_out/pmc_scatter/summary.txt
