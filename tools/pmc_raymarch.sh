#!/bin/bash
# usage: tools/pmc_raymarch.sh <tag>  -- separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only;
# writes gpurun_out/pmc/raymarch_pmc.json (copy to profiles/raymarch_pmc.json: bench.py's roofline.traffic reads it)
tag=${1:-untagged}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python tools/pmc_raymarch.py > gpurun_out/pmc/$c.log 2>&1 < /dev/null
  echo "$c rc=$?"
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "raymarch|Counter_Name|Kernel_Name" "$f" | head -30 > gpurun_out/pmc/${c}_raymarch.csv; fi
done
python tools/pmc_raymarch_json.py gpurun_out/pmc/FETCH_SIZE_raymarch.csv gpurun_out/pmc/WRITE_SIZE_raymarch.csv "$tag" > gpurun_out/pmc/raymarch_pmc.json
cat gpurun_out/pmc/raymarch_pmc.json
