#!/bin/bash
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python tools/pmc_raymarch.py > gpurun_out/pmc/$c.log 2>&1 < /dev/null
  echo "$c rc=$?"
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "raymarch|Counter_Name|Kernel_Name" "$f" | head -30 > gpurun_out/pmc/${c}_raymarch.csv; fi
done
head -3 gpurun_out/pmc/FETCH_SIZE_raymarch.csv | cut -c1-400
