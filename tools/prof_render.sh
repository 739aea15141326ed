#!/bin/bash
# usage: tools/prof_render.sh <tag>  -- rocprofv3 kernel-trace statistics of one config-size render forward + backward (tools/pmc_render.py)
tag=${1:-render}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels --output-format csv -d /tmp/profr_$tag -o $tag -- python tools/pmc_render.py > gpurun_out/prof_render_$tag.log 2>&1 < /dev/null
echo "rocprof rc=$?"
f=$(find /tmp/profr_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_render_kernel_stats.csv && head -25 "$f"
