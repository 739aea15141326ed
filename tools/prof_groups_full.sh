#!/bin/bash
# usage: prof_groups_full.sh <tag> <bench args...>  -- per-(kernel with template arguments, grid) totals of bench.py (rocprofv3 kernel trace, names not truncated)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/groups_$tag
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/groups_$tag -o $tag -- python bench.py --no-cpu-baseline --no-dense-leg --alt-conv-precision none "$@" > gpurun_out/groups_$tag/bench.log 2>&1 < /dev/null
echo "rocprof rc=$?"
tr=$(find /tmp/groups_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$tr" ] && python tools/trace_groups.py "$tr" 400 > gpurun_out/groups_$tag/trace_groups_full.txt
head -60 gpurun_out/groups_$tag/trace_groups_full.txt
