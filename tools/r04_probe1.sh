#!/bin/bash
# r04_probe1.sh -- round 4, first GPU call: the step time by calls in flight x trellis kernel x hardware queues, and SQ counters of the
# final round-3 kernels alone (4096 frames) and at saturation (16384 frames per launch).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
Q="--no-cpu-baseline --no-extras --check 64"
: > $OUT/r04_a_depth_table.txt
for cfg in "0 1 16" "0 1 64" "0 2 16" "0 2 64" "0 3 64" "0 3 16" "0 4 16" "0 8 16" "16 8 16"; do
  set -- $cfg
  line=$(timeout 300 python bench.py $Q --hw-queues $1 --depth $2 --trellis $3 2>/dev/null | tail -1)
  echo "hwq=$1 depth=$2 trellis=$3 $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "value", d["value"], "hw_queues", d["config"].get("hw_queues"), "kernel_ms", d.get("kernel_ms"), "alone", d.get("kernel_ms_one_call_in_flight"), "lat", d.get("realtime",{}).get("call_latency_ms_one_in_flight"))' 2>&1)" >> $OUT/r04_a_depth_table.txt
done
cat $OUT/r04_a_depth_table.txt
bash tools/pmc_sq.sh r04_a_alone --trellis 16 > $OUT/r04_a_sq_alone.log 2>&1
bash tools/pmc_sq.sh r04_a_sat --trellis 16 --frames 16384 > $OUT/r04_a_sq_sat.log 2>&1
tail -5 $OUT/r04_a_sq_alone.log
