#!/bin/bash
# prof_11n.sh <tag> -- on the GPU box: rocprofv3 kernel stats of the 802.11n bench section (staged chain), copied to gpurun_out/<tag>_11n_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-r02}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_11n -o p -- python $R/bench.py --no-cpu-baseline --only rx11n > $OUT/${TAG}_11n_bench.json 2> $OUT/${TAG}_11n.err
K=$(find $OUT/${TAG}_11n -name "*kernel_stats.csv" | head -1); cp "$K" $OUT/${TAG}_11n_kernel_stats.csv; rm -rf $OUT/${TAG}_11n
head -8 $OUT/${TAG}_11n_kernel_stats.csv
