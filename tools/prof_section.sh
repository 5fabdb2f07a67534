#!/bin/bash
# prof_section.sh <tag> <bench key> -- on the GPU box: rocprofv3 kernel stats of one bench section (bench.py --only <key>) -> gpurun_out/<tag>_<key>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${1:-r03}; KEY=${2:-rx11b_cck}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${KEY} -o p -- python $R/bench.py --no-cpu-baseline --only $KEY > $OUT/${TAG}_${KEY}_bench.json 2> $OUT/${TAG}_${KEY}.err
K=$(find $OUT/${TAG}_${KEY} -name "*kernel_stats.csv" | head -1); cp "$K" $OUT/${TAG}_${KEY}_kernel_stats.csv; rm -rf $OUT/${TAG}_${KEY}
head -8 $OUT/${TAG}_${KEY}_kernel_stats.csv | cut -c1-200
