#!/bin/bash
# separate --pmc passes over one config-size render forward + backward (kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_render
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcr_$i -o pmc -- python tools/pmc_render.py > gpurun_out/pmc_render/pass$i.log 2>&1 < /dev/null
  echo "pass $i rc=$?"
done
python tools/pmc_summarize.py /tmp/pmcr_1 /tmp/pmcr_2 /tmp/pmcr_3 /tmp/pmcr_4 | grep -E "decode|raymarch|merge_sort|importance|plane_scatter|bin_points" | tee gpurun_out/pmc_render/summary.txt
