#!/bin/bash
# collect_fused_traffic.sh <tag> -- on the GPU box: the PMC passes of collect_profiles.sh with the data field decoded by k_decode (SORA_HIP_FUSED=1)
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SORA_HIP_FUSED=1
PMC="--no-cpu-baseline --no-extras --check 64 --steps 3 --warmup 1 --depth 1 --min-seconds 0 --no-deliver"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_ffetch -o p -- python $R/bench.py $PMC > /dev/null 2> $OUT/${TAG}_ffetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_fwrite -o p -- python $R/bench.py $PMC > /dev/null 2> $OUT/${TAG}_fwrite.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_finsts -o p -- python $R/bench.py $PMC > /dev/null 2> $OUT/${TAG}_finsts.err
I=$(find $OUT/${TAG}_finsts -name "*counter_collection.csv" | head -1)
F=$(find $OUT/${TAG}_ffetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/${TAG}_fwrite -name "*counter_collection.csv" | head -1)
python $R/tools/summarize_pmc.py "$F" "$W" 4096 "$I" > $OUT/${TAG}_traffic_fused.json
rm -rf $OUT/${TAG}_ffetch $OUT/${TAG}_fwrite $OUT/${TAG}_finsts
cat $OUT/${TAG}_traffic_fused.json
