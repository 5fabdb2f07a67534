#!/bin/bash
# collect_profiles.sh <tag> -- run on the GPU box (gpurun): kernel-trace stats of the headline loop (the handle's default calls in flight -- under the
# tracer the host is the bottleneck there -- and once more with ONE call in flight, so that every kernel's duration is the kernel alone on the chip) + the
# PMC passes (their own runs, one call in flight so that every dispatch is attributed cleanly).
# Outputs land in gpurun_out/<tag>_*; copy the summaries to profiles/ afterwards (see DESIGN.md, section "Measurement").
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-extras --no-plain --check 64"   # (--no-plain: the plain_host section launches 32768-capture calls, which do not belong in per-launch means of the 4096-capture call)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -o p -- python $R/bench.py $QUICK --headline-only --min-seconds 2 > $OUT/${TAG}_ks_bench.json 2> $OUT/${TAG}_ks.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks1 -o p -- python $R/bench.py $QUICK --headline-only --min-seconds 2 --depth 1 --trellis 16 > $OUT/${TAG}_ks1_bench.json 2> $OUT/${TAG}_ks1.err
# the PMC passes, once per trellis kernel (sora_rx_set_trellis: 16 = k_viterbi16, what the default bench uses; 64 = k_viterbi; 1 = the window-parallel k_viterbi16w)
for T in ${TRELLISES:-16 64 1}; do
  PMC="$QUICK --steps 3 --warmup 1 --depth 1 --trellis $T --min-seconds 0 --no-deliver"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o p -- python $R/bench.py $PMC > /dev/null 2> $OUT/${TAG}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o p -- python $R/bench.py $PMC > /dev/null 2> $OUT/${TAG}_write.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/${TAG}_insts -o p -- python $R/bench.py $PMC > /dev/null 2> $OUT/${TAG}_insts.err
  I=$(find $OUT/${TAG}_insts -name "*counter_collection.csv" | head -1)
  F=$(find $OUT/${TAG}_fetch -name "*counter_collection.csv" | head -1)
  W=$(find $OUT/${TAG}_write -name "*counter_collection.csv" | head -1)
  if [ $T = 16 ]; then python $R/tools/summarize_pmc.py "$F" "$W" 4096 "$I" k_viterbi16 > $OUT/${TAG}_traffic.json;
  elif [ $T = 64 ]; then python $R/tools/summarize_pmc.py "$F" "$W" 4096 "$I" k_viterbi > $OUT/${TAG}_traffic_trellis64.json;
  else python $R/tools/summarize_pmc.py "$F" "$W" 4096 "$I" k_viterbi16w > $OUT/${TAG}_traffic_windowed.json; fi     # (the window-parallel form: k_viterbi16w + k_win_redo)
  rm -rf $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_insts
done
K=$(find $OUT/${TAG}_ks -name "*kernel_stats.csv" | head -1)
cp "$K" $OUT/${TAG}_kernel_stats.csv
K1=$(find $OUT/${TAG}_ks1 -name "*kernel_stats.csv" | head -1)
cp "$K1" $OUT/${TAG}_kernel_stats_one_call_in_flight.csv
rm -rf $OUT/${TAG}_ks $OUT/${TAG}_ks1
tail -1 $OUT/${TAG}_ks_bench.json | head -c 1500
cat $OUT/${TAG}_traffic.json
head -8 $OUT/${TAG}_kernel_stats.csv
head -8 $OUT/${TAG}_kernel_stats_one_call_in_flight.csv
