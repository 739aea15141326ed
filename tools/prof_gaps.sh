#!/bin/bash
# usage: prof_gaps.sh <tag> <window_ms> <bench args...>  -- rocprofv3 kernel trace of bench.py, GPU idle-gap analysis of the last <window_ms> of it
tag=$1; win=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gaps_$tag
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_$tag -o $tag -- python bench.py --no-cpu-baseline --no-dense-leg --alt-conv-precision none "$@" > gpurun_out/gaps_$tag/bench.log 2>&1 < /dev/null
echo "rocprof rc=$?"
tr=$(find /tmp/gaps_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$tr" ] && python tools/trace_gaps.py "$tr" $win | tee gpurun_out/gaps_$tag/gaps.txt
grep -E "^\{" gpurun_out/gaps_$tag/bench.log | cut -c1-400
