#!/bin/bash
# prof_row.sh <row> : kernel-trace stats of one bench row
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pr_$1 -o p -- python $R/bench.py --no-cpu-baseline --only $1 > $OUT/pr_$1.json 2> $OUT/pr_$1.err
K=$(find $OUT/pr_$1 -name "*kernel_stats.csv" | head -1); cp "$K" $OUT/pr_$1_kernel_stats.csv; rm -rf $OUT/pr_$1
head -14 $OUT/pr_$1_kernel_stats.csv | cut -c1-200
