#!/bin/bash
# trace_timeline.sh <tag> <trellis> <depth> (TL_ARGS = more bench flags, e.g. "--frames 32768 --no-plain") -- on the GPU box: rocprofv3 kernel trace of the headline loop, kept as CSV for tools/timeline.py
TAG=$1; T=$2; D=$3
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_kt -o t -- python $R/bench.py --no-cpu-baseline --no-extras --depth $D --trellis $T --check 64 --headline-only --steps 40 --warmup 10 --min-seconds 0.3 ${TL_ARGS:-} > $OUT/${TAG}_kt.json 2> $OUT/${TAG}_kt.err
find $OUT/${TAG}_kt -name "*kernel_trace.csv" -exec cp {} $OUT/${TAG}_kernel_trace.csv \;
rm -rf $OUT/${TAG}_kt
wc -l $OUT/${TAG}_kernel_trace.csv
