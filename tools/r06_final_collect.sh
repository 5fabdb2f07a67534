#!/bin/bash
# r06_final_collect.sh -- on the GPU box: everything profiles/r06_final_* is made of, in one go (about ten minutes).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 bash $R/tools/collect_profiles.sh r06_final > $OUT/r06_final_collect.log 2>&1
cp $OUT/r06_final_traffic*.json $R/profiles/ 2>/dev/null        # (what the bench runs below replay: stamped with this tree's sources)
cd /tmp && export TMPDIR=/tmp
# the shard shape's kernels (BASELINE configs[4]'s per-GPU share, one call in flight) and a lone capture's (fsample-6, the library's automatic choice)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_shard_ks -o p -- python $R/bench.py --shape shard --no-cpu-baseline --headline-only --min-seconds 1 --depth 1 > $OUT/r06_shard_ks_bench.json 2> $OUT/r06_shard_ks.err
K=$(find $OUT/r06_shard_ks -name "*kernel_stats.csv" | head -1); cp "$K" $OUT/r06_final_kernel_stats_shard_32x16_one_call_in_flight.csv; rm -rf $OUT/r06_shard_ks
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/r06_single -o p -- python $R/tools/r05_window_latency.py --profile-single 30 > $OUT/r06_single.log 2>&1
DB=$(find $OUT/r06_single -name "*_results.db" | head -1)
python $R/tools/rocprof_kernels.py "$DB" --last 3 > $OUT/r06_single_capture_kernels.json 2>> $OUT/r06_single.log
rm -rf $OUT/r06_single
cd $R
for a in "--shard" "--fsample6" ""; do echo "== probe_scan $a"; timeout 200 python tools/probe_scan.py $a 2>&1 | tail -9; done > $OUT/r06_probe_scan.txt
timeout 600 python bench.py --shape shard --no-cpu-baseline > $OUT/r06_final_bench_shape_shard.json 2> $OUT/r06_final_bench_shape_shard.err
timeout 1200 python bench.py > $OUT/r06_final_bench.json 2> $OUT/r06_final_bench.err
tail -c 300 $OUT/r06_final_bench.json; tail -3 $OUT/r06_final_bench.err; cat $OUT/r06_probe_scan.txt | tail -30; head -12 $OUT/r06_final_kernel_stats_shard_32x16_one_call_in_flight.csv
