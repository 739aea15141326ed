#!/bin/bash
# usage: prof_markers.sh <tag> [bench args]  -- rocprofv3 --marker-trace --kernel-trace of bench.py with SPI_TRACE=1 (rocTX ranges at the
# reference's profiled_function boundaries and around the phases of both loops, torch_utils/misc.py); eager steps so every range exists
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
SPI_TRACE=1 SPI_STAGE1_GRAPH=0 timeout 900 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --no-cpu-baseline --alt-conv-precision none "$@" > gpurun_out/prof_$tag/bench.log 2>&1 < /dev/null
echo "rocprof rc=$?"
python tools/marker_groups.py /tmp/prof_$tag > gpurun_out/prof_$tag/marker_ranges.txt
head -40 gpurun_out/prof_$tag/marker_ranges.txt
