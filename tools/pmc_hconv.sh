#!/bin/bash
# kernel trace + PMC passes for the direct fp16 conv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SPI_BENCH_F16=1 SPI_BENCH_HALF=1
mkdir -p $R/gpurun_out/hc
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/hc/kt -o kt --output-format csv -- python $R/tools/pmc_conv.py > /dev/null 2>&1
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp -d $R/gpurun_out/hc/pmc_$tag -o p --output-format csv -- python $R/tools/pmc_conv.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for f in glob.glob(R+'/gpurun_out/hc/kt/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        print(row['Name'][:70], row['Calls'], row['AverageNs'])
acc=collections.defaultdict(list)
for f in glob.glob(R+'/gpurun_out/hc/pmc_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        for kn in ('hconv_kernel', 'hwgrad_kernel', 'hwgrad_reduce_kernel', 'hconv_weight_kernel'):
            if kn + '(' in row['Kernel_Name'] or row['Kernel_Name'].startswith('_Z%d%s' % (len(kn), kn)):
                acc[(kn, row['Counter_Name'])].append(float(row['Counter_Value']))
for (kn, k), v in sorted(acc.items()):
    print(f'{kn:22s} {k:32s} {sum(v)/len(v):14.1f} n={len(v)}')
PY
