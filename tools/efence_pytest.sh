#!/bin/bash
# Bounds-checked pass over the GPU suite: tests/conftest.py installs the guard-page allocator (tools/efence/efence_alloc.cpp) when SPI_EFENCE=1.
# One pytest process per test file (a GPU memory fault aborts the process: the summary names the file and the last test that started).
#   gpurun -- 'bash tools/efence_pytest.sh [per-file timeout s] [file ...]'  ->  gpurun_out/efence/summary.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=${1:-900}; shift
FILES=${@:-$(ls tests/test_hip_*_gpu.py)}
mkdir -p gpurun_out/efence
hipcc -O2 -shared -fPIC -o tools/efence/libefence.so tools/efence/efence_alloc.cpp || exit 1
: > gpurun_out/efence/summary.txt
for f in $FILES; do
  b=$(basename $f .py)
  # (graph-replay and arena tests are excluded: the pass runs every loop eagerly and every accumulator as its own allocation, see conftest.py)
  SPI_EFENCE=1 timeout $T python -m pytest $f -m gpu -x -q -v -k "not graph and not zero_arena" -p no:cacheprovider > gpurun_out/efence/$b.log 2>&1
  rc=$?
  last=$(grep -E "PASSED|FAILED|ERROR" gpurun_out/efence/$b.log | tail -1 | cut -c1-120)
  res=$(grep -E "passed|failed|error" gpurun_out/efence/$b.log | tail -1)
  fault=$(grep -i -m1 "memory access fault\|Memory access fault\|page not present" gpurun_out/efence/$b.log)
  echo "$b rc=$rc | $res | last: $last | ${fault:-no GPU fault}" | tee -a gpurun_out/efence/summary.txt
done
grep -h "\[efence\]" gpurun_out/efence/*.log | sort | uniq -c | head -5 >> gpurun_out/efence/summary.txt
