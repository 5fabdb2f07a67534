#!/bin/bash
# collect_profiles.sh <tag> -- run on the GPU box (gpurun): kernel-trace stats of the bench command + the two PMC passes.
# Outputs land in gpurun_out/<tag>_*; copy the summaries to profiles/ afterwards (see DESIGN.md, section "Measurement").
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -o p -- python $R/bench.py --no-cpu-baseline > $OUT/${TAG}_ks_bench.json 2> $OUT/${TAG}_ks.err
# counters: their own passes, one call in flight so that every dispatch is attributed cleanly
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --depth 1 --check 0 > /dev/null 2> $OUT/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --depth 1 --check 0 > /dev/null 2> $OUT/${TAG}_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_insts -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --depth 1 --check 0 > /dev/null 2> $OUT/${TAG}_insts.err
I=$(find $OUT/${TAG}_insts -name "*counter_collection.csv" | head -1)
F=$(find $OUT/${TAG}_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/${TAG}_write -name "*counter_collection.csv" | head -1)
python $R/tools/summarize_pmc.py "$F" "$W" 4096 "$I" > $OUT/${TAG}_traffic.json
K=$(find $OUT/${TAG}_ks -name "*kernel_stats.csv" | head -1)
cp "$K" $OUT/${TAG}_kernel_stats.csv
tail -1 $OUT/${TAG}_ks_bench.json
cat $OUT/${TAG}_traffic.json
head -8 $OUT/${TAG}_kernel_stats.csv
