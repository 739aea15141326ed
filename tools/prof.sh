#!/bin/bash
# usage: tools_prof.sh <tag> <bench args...>   -- rocprofv3 kernel stats of bench.py, keeps only the small summaries
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --truncate-kernels --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py "$@" > gpurun_out/prof_$tag/bench.log 2>&1 < /dev/null
echo "rocprof rc=$?"
find /tmp/prof_$tag -type f -size -2M \( -name "*stats*" -o -name "*agent*" \) -exec cp {} gpurun_out/prof_$tag/ \;
tr=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$tr" ] && python tools/trace_groups.py "$tr" 70 > gpurun_out/prof_$tag/trace_groups.txt
grep -E "^\{" gpurun_out/prof_$tag/bench.log | cut -c1-900
