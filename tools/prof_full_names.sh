#!/bin/bash
# usage: prof_full_names.sh <tag>  -- rocprofv3 kernel stats of the default bench.py WITHOUT --truncate-kernels, so template
# instances stay apart (raymarch_fwd_kernel<3> = the final S=192 composite that bench.py's roofline object times with HIP events)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --no-cpu-baseline "$@" > gpurun_out/prof_$tag/bench.log 2>&1 < /dev/null
echo "rocprof rc=$?"
st=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$st" ] && { head -1 "$st"; grep -E "raymarch|decode_|igemm_kernel<2, 2, 2, 2|wgrad_kernel" "$st" | head -40; } > gpurun_out/prof_$tag/kernel_stats_full_names.csv
grep -E "^\{" gpurun_out/prof_$tag/bench.log | cut -c1-300
